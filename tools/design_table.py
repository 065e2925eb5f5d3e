#!/usr/bin/env python3
"""Prints DESIGN.md section 5's measurement table from a tracked bench line (profiles/rNN_bench_1gpu.json)."""
import json, sys
d = json.loads(open(sys.argv[1] if len(sys.argv) > 1 else "profiles/r02_bench_1gpu.json").read().strip().splitlines()[-1])
v, r, cb = d["variants_1gpu"], d["roofline"], d["cpu_baseline"]
hp = v["host_path_pcie_inclusive_1x4K_parity"]


def row(name, ms, gbs, frac, mpix, r1):
    return f"| {name} | {ms} | {gbs} | {frac} | {mpix} | {r1} |"


def var(key, name, r1="", tkey="kernel_ms_avg", note=""):
    x = v[key]
    return row(name, f"{x[tkey]*1e3:.1f} µs", f"{x['achieved_GBs']:.0f}", f"{x['frac']*100:.1f} %{note}", f"{x['Mpixels_per_s']:,.0f}", r1)


print("| | time / step | algorithmic GB/s | of 8 TB/s | Mpix/s | driver, round 1 |\n|---|---|---|---|---|---|")
print(row("PARITY border 40 (headline)", f"{r['kernel_ms_avg']*1e3:.1f} µs", f"{r['achieved']:.0f}", f"**{r['frac']*100:.1f} %**",
          f"{d['value']:,.0f}", "435.7 µs, 71.8 %"))
print(var("parity_border0", "PARITY border 0"))
print(var("parity_u8_input_border40", "PARITY, u8 input (fused cpp:61 decode, 1+16 B/px)", "376.2 µs"))
print(var("compact_border40_all_valid", "COMPACT all-valid, border 40 (single pass)", "501.9 µs, 62.3 %"))
print(var("compact_border40_30pct_holes_index", "COMPACT 30 % holes + indices (single pass)", "567.8 µs, 49.6 %"))
print(var("compact_1080p_30pct_holes_index_32frames", "COMPACT, config 3's geometry: 32 × 1920×1080, 30 % holes + indices (single pass)", "", "ms_per_launch"))
print(var("compact_1080p_30pct_holes_index_1frame", "COMPACT, config 3's geometry: ONE 1920×1080 frame (count + self-scanning scatter)", "", "ms_per_launch", " (launch latency)"))
x = v["callback_u8_median11_parity_border40"]
print(row("callback body `d2pc_process_mono_device`: ROI median 11×11 + reproject (u8 in), ONE kernel (`k_callback_bs`)",
          f"{x['kernel_ms_avg']*1e3:.1f} µs (as two launches with the bit-sliced filter {x['as_two_launches_ms']*1e3:.1f}; with round 1's filter 1031–1103)",
          "—", "—", f"{x['Mpixels_per_s']:,.0f}", "1252.6 µs"))
print(row("host path, one 4K frame incl. PCIe: `d2pc_process` pageable / pinned frame + cloud / pipelined direct",
          f"{hp['sync_d2pc_process']['ms_per_frame']:.2f} / {hp['sync_d2pc_process_pinned_io']['ms_per_frame']:.2f} / {hp['pipelined_direct_host_write']['ms_per_frame']:.2f} ms",
          "—", "—", f"{hp['sync_d2pc_process']['Mpixels_per_s']:,.0f} / {hp['sync_d2pc_process_pinned_io']['Mpixels_per_s']:,.0f} / {hp['pipelined_direct_host_write']['Mpixels_per_s']:,.0f}", ""))
print(row("CPU oracle, 1 thread (reference-like; gcc -O3 -march=x86-64-v3)", "—", "—", "—",
          f"{cb['value']:.0f} ({cb['all_cores']['cores']} threads: {cb['all_cores']['value']:.0f})", "297 (2,205)"))
print(f"\nbuild {d['config']['build']}, traffic profile build {r['traffic_profile_build']}")
