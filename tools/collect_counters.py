#!/usr/bin/env python3
"""Condense the passes of tools/pmc_passes.sh (gpurun_out/pmc_<tag>/<pass>/...) into ONE tracked json:
per kernel and counter, the mean over dispatches (first dispatch = warm-up, dropped), plus the ratios
the design discussion uses.  Usage: collect_counters.py <out.json> <tag>[=label] ..."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(tag):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    durs = collections.defaultdict(list)
    logs = {}
    for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}", "*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows = collections.defaultdict(lambda: collections.defaultdict(dict))
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if "d2pc::" not in k:
                    continue
                rows[k][r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
            for k, cs in rows.items():
                for c, byd in cs.items():
                    vals = [byd[i] for i in sorted(byd)][1:] or list(byd.values())
                    per[k][c].extend(vals)
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if "d2pc::" in k:
                    durs[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        lg = d + ".log"
        if os.path.exists(lg):
            for line in open(lg):
                if line.startswith("{"):
                    logs[os.path.basename(d)] = json.loads(line)
    out = {}
    for k, cs in per.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        m["dispatches_averaged"] = max(len(v) for v in cs.values())
        if durs.get(k):
            m["duration_us_under_pmc_mean"] = sum(durs[k]) / len(durs[k])
        out[k] = m
    return out, logs


def derive(m, probe):
    d = {}
    g = m.get
    if g("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM",
                  "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA"):
            if g(c) is not None and c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                d[c + "/SQ_WAVE_CYCLES"] = g(c) / g("SQ_WAVE_CYCLES")
    if g("SQ_INSTS_VALU") and g("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; a wave64 VALU instruction occupies its SIMD-32 for 2 cycles (4 for fp64
        # and a few integer forms): the fraction below is therefore a LOWER bound of the vector-issue slots taken
        cyc = g("GRBM_GUI_ACTIVE") / 8.0
        d["gpu_cycles_per_launch(GRBM_GUI_ACTIVE/8)"] = cyc
        d["valu_issue_frac(SQ_INSTS_VALU x 2 cycles / 1024 SIMDs / cycles)"] = g("SQ_INSTS_VALU") * 2.0 / 1024.0 / cyc
        d["simd_cycles_per_valu_wave_instruction"] = 1024.0 * cyc / g("SQ_INSTS_VALU")
        if g("duration_us_under_pmc_mean"):
            d["effective_clock_ghz_under_pmc"] = cyc / g("duration_us_under_pmc_mean") / 1e3
        if g("SQ_LDS_IDX_ACTIVE") is not None:
            d["lds_array_busy_frac(SQ_LDS_IDX_ACTIVE / 256 CUs / cycles)"] = g("SQ_LDS_IDX_ACTIVE") / 256.0 / cyc
            d["lds_bank_conflict_frac_of_lds_cycles"] = g("SQ_LDS_BANK_CONFLICT", 0.0) / max(g("SQ_LDS_IDX_ACTIVE"), 1.0)
        if g("SQ_WAVE_CYCLES"):
            d["waves_resident_per_cu(SQ_WAVE_CYCLES x 4 / cycles / 256)"] = g("SQ_WAVE_CYCLES") * 4.0 / cyc / 256.0
    if probe and g("SQ_INSTS_VMEM_WR"):
        px = probe["roi_pixels"]
        d["pixels_per_launch"] = px
        d["points_per_launch"] = probe["points"]
        for c in ("SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"):
            if g(c) is not None:
                d[c + "_per_1024_pixels(wave-instructions)"] = g(c) / px * 1024
    if g("TCP_TCC_WRITE_REQ") and probe:
        d["TCP_TCC_WRITE_REQ_x64B/bytes_written"] = g("TCP_TCC_WRITE_REQ") * 64 / ((20 if "idx" in probe["workload"] or "1080p" in probe["workload"] else 16) * probe["points"])
    if g("TCP_TCC_WRITE_REQ_LATENCY") and g("TCP_TCC_WRITE_REQ"):
        d["write_round_trip_cycles"] = g("TCP_TCC_WRITE_REQ_LATENCY") / g("TCP_TCC_WRITE_REQ")
    if g("TCP_TCC_READ_REQ_LATENCY") and g("TCP_TCC_READ_REQ"):
        d["read_round_trip_cycles"] = g("TCP_TCC_READ_REQ_LATENCY") / g("TCP_TCC_READ_REQ")
    if g("TCP_PENDING_STALL_CYCLES") and g("TCP_GATE_EN1"):
        d["TCP_PENDING_STALL_CYCLES/TCP_GATE_EN1"] = g("TCP_PENDING_STALL_CYCLES") / g("TCP_GATE_EN1")
    if g("TCC_EA0_WRREQ"):
        d["TCC_EA0_WRREQ_64B/TCC_EA0_WRREQ"] = g("TCC_EA0_WRREQ_64B", 0) / g("TCC_EA0_WRREQ")
        if g("TCC_EA0_WRREQ_STALL") is not None and g("TCC_CYCLE"):
            d["TCC_EA0_WRREQ_STALL/TCC_CYCLE"] = g("TCC_EA0_WRREQ_STALL") / g("TCC_CYCLE")
    if g("TCC_HIT") is not None and g("TCC_MISS") is not None and g("TCC_HIT") + g("TCC_MISS") > 0:
        d["L2_hit_rate"] = g("TCC_HIT") / (g("TCC_HIT") + g("TCC_MISS"))
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        d["hbm_read_bytes(FETCH_SIZE KiB x1024 x2 gfx950)"] = g("FETCH_SIZE") * 1024 * 2
        d["hbm_write_bytes(WRITE_SIZE KiB x1024)"] = g("WRITE_SIZE") * 1024
        if probe:
            d["traffic/algorithmic"] = (g("FETCH_SIZE") * 2048 + g("WRITE_SIZE") * 1024) / probe["algorithmic_bytes"]
    return d


def main():
    out_path = sys.argv[1]
    res = {"how": "tools/pmc_passes.sh <workload> <tag>: separate rocprofv3 --pmc passes of tools/probe.py, kernel-trace only; "
                  "means over dispatches 2..N of each pass; SQ_* cycle counters are in quad-cycles summed over waves/SEs",
           "workloads": {}}
    for spec in sys.argv[2:]:
        tag, _, label = spec.partition("=")
        per, logs = load(tag)
        probe = next(iter(logs.values()), None)
        entry = {"probe_line_of_one_pass": probe, "kernels": {}}
        for k, m in per.items():
            entry["kernels"][k] = {"counters": m, "derived": derive(m, probe)}
        res["workloads"][label or tag] = entry
    json.dump(res, open(out_path, "w"), indent=1)
    for w, e in res["workloads"].items():
        for k, v in e["kernels"].items():
            print(w, k[:60])
            for a, b in v["derived"].items():
                print("   ", a, round(b, 4) if isinstance(b, float) else b)


if __name__ == "__main__":
    main()
