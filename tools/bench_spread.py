#!/usr/bin/env python3
"""Condenses several bench.py JSON lines (one per gpurun call = one device draw) into the table kept as
profiles/rNN_bench_spread.txt.   python tools/bench_spread.py a.json b.json c.json > profiles/r03_bench_spread.txt"""
import json, sys

runs = []
for p in sys.argv[1:]:
    with open(p) as f:
        line = [l for l in f.read().splitlines() if l.startswith("{")][-1]
    runs.append(json.loads(line))
if not runs:
    sys.exit("usage: bench_spread.py bench_a.json [bench_b.json ...]")
builds = sorted({r["config"].get("build", "?") for r in runs})
print(f"# python bench.py (defaults: --steps {runs[0]['steps']} --warmup {runs[0]['warmup']}), {len(runs)} gpurun calls = {len(runs)} device draws, "
      f"build {', '.join(builds)} (inputs: {', '.join(p.split('/')[-1] for p in sys.argv[1:])}).")
print("# per variant: median ms per launch over rounds that sum to >= 100 ms (fraction of 8 TB/s); 'two:' = the same work as two launches")
print("headline: contract ms/step | frac | sustained median ms (frac) | device fill / copy GB/s, persistent blocks | one-shot blocks (plain, nt)")
cells = []
for r in runs:
    ro = r["roofline"]
    sp = ro.get("kernel_ms_spread", {})
    fr = ro.get("frac_at_min_median_max_ms", [None, None, None])
    cells.append(f"{r['ms_per_step']} | {ro['frac']} | {sp.get('median')} ({fr[1]}) | {ro.get('device_fill_GBs')} / {ro.get('device_copy_GBs')} | "
                 f"{ro.get('device_fill_oneshot_plain_GBs')}, {ro.get('device_fill_oneshot_nt_GBs')} / {ro.get('device_copy_oneshot_plain_GBs')}, {ro.get('device_copy_oneshot_nt_GBs')}")
print("    " + "  ||  ".join(cells))
names = []
for r in runs:
    for k in r.get("variants_1gpu", {}):
        if k not in names:
            names.append(k)
for k in names:
    if k.startswith("host_path"):
        continue
    print(k)
    cells = []
    for r in runs:
        v = r.get("variants_1gpu", {}).get(k)
        if not v:
            cells.append("-")
            continue
        ms = (v.get("kernel_ms_spread") or v.get("ms_per_launch_spread") or {}).get("median", v.get("kernel_ms_avg", v.get("ms_per_launch")))
        two = v.get("as_two_launches_ms_spread")
        cells.append(f"{ms} ({v.get('frac')})" + (f" two:{two['median']}" if two else ""))
    print("    " + "  ||  ".join(cells))
hp = [k for k in names if k.startswith("host_path")]
for k in hp:
    print(k + " (ms per 4K frame: pageable / pinned / pipelined)")
    cells = []
    for r in runs:
        v = r["variants_1gpu"].get(k, {})
        cells.append("/".join(str(v.get(n, {}).get("ms_per_frame")) for n in ("sync_d2pc_process", "sync_d2pc_process_pinned_io", "pipelined_direct_host_write")))
    print("    " + "  ||  ".join(cells))
print("cpu_baseline Mpixel/s: 1 thread / all cores")
cells = []
for r in runs:
    c = r.get("cpu_baseline") or {}
    allc = c.get("all_cores") or {}
    cells.append(f"{c.get('value')} / {allc.get('value')} ({allc.get('cores')})")
print("    " + "  ||  ".join(cells))
