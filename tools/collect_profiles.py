#!/usr/bin/env python3
"""Condense the rocprofv3 output of one round (gpurun_out/prof_rNN_*) into the
tracked files under profiles/:
  rNN_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary (verbatim)
  rNN_pmc_summary.json      FETCH_SIZE / WRITE_SIZE per kernel (+ calibration)
  hbm_traffic.json          HBM bytes per launch, read by bench.py ("traffic")
Corrections follow MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are
in KiB; on gfx950 FETCH_SIZE tallies a 128-B request as 64 B, so it is doubled
-- confirmed here for THIS kernel's load shape by the calibration pass
(tools/membench calib: known byte counts read with dword and dwordx4 loads).
"""
import collections, csv, glob, hashlib, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
num = rnd[1:].lstrip("0") or "0"
tag = f"prof_r{num}"

def build_id():  # the same id bench.py puts into config.build
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "disparity_to_point_cloud_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:12]

def one(pattern):
    fs = sorted(glob.glob(os.path.join(src, pattern.replace("/*/", "/**/")), recursive=True))
    return fs[-1] if fs else None

stats = one(f"{tag}_stats/*/*_kernel_stats.csv")
if stats:  # the headline-only run (bench.py --no-variants): k_reproject_pack<F32> = the border-40 workload alone
    shutil.copy(stats, os.path.join(dst, f"{rnd}_kernel_stats.csv"))
stats_v = one(f"{tag}_stats_variants/*/*_kernel_stats.csv")
if stats_v:  # the run with all side measurements: every other kernel of the library
    shutil.copy(stats_v, os.path.join(dst, f"{rnd}_kernel_stats_variants.csv"))
for suffix in ("under_rocprof", "under_rocprof_variants"):
    b = os.path.join(src, f"bench_r{num}_{suffix}.json")
    if os.path.exists(b) and os.path.getsize(b) > 0:
        shutil.copy(b, os.path.join(dst, f"{rnd}_bench_{suffix}.json"))

def counters(dirname):
    out = collections.defaultdict(list)
    f = one(f"{dirname}/*/*_counter_collection.csv")
    if not f:
        return out
    for r in csv.DictReader(open(f)):
        out[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return out

summary = {"units": "FETCH_SIZE/WRITE_SIZE in KiB as reported; *_bytes corrected", "kernels": {}, "calibration": {}}
cal_f, cal_w = counters(f"{tag}_calib_fetch"), counters(f"{tag}_calib_write")
known = 16 * 3840 * 2160 * 16  # bytes per membench calib launch
fetch_factor = None
for (k, c), v in cal_f.items():
    if k.endswith("k_read4") and c == "FETCH_SIZE":
        fetch_factor = known / (sum(v) / len(v) * 1024)
        summary["calibration"]["k_read4 (dword loads, known %d B)" % known] = {"FETCH_SIZE_KiB": sum(v) / len(v), "true/reported": fetch_factor}
    if "k_read<" in k and c == "FETCH_SIZE":
        summary["calibration"]["k_read (dwordx4 loads, known %d B)" % known] = {"FETCH_SIZE_KiB": sum(v) / len(v), "true/reported": known / (sum(v) / len(v) * 1024)}
for (k, c), v in cal_w.items():
    if "k_fill" in k and c == "WRITE_SIZE":
        summary["calibration"]["k_fill (dwordx4 stores, known %d B)" % known] = {"WRITE_SIZE_KiB": sum(v) / len(v), "true/reported": known / (sum(v) / len(v) * 1024)}
if fetch_factor is None:
    fetch_factor = 2.0
fetch, write = counters(f"{tag}_fetch"), counters(f"{tag}_write")
for extra in (f"{tag}_cfetch", f"{tag}_cwrite"):  # the same passes with bench.py --mode compact
    for k, v in counters(extra).items():
        (fetch if k[1] == "FETCH_SIZE" else write).setdefault(k, []).extend(v)
sq = counters(f"{tag}_sq")
if sq:
    summary["sq_counters_parity_kernel"] = {c: sum(v) / len(v) for (k, c), v in sq.items() if "k_reproject_pack" in k}
per_kernel = collections.defaultdict(dict)
for (k, c), v in list(fetch.items()) + list(write.items()):
    if "d2pc::" in k:
        per_kernel[k][c] = sum(v) / len(v)
        per_kernel[k]["launches_" + c] = len(v)
traffic = {}
for k, d in per_kernel.items():
    d["read_bytes"] = d.get("FETCH_SIZE", 0) * 1024 * round(fetch_factor, 3)
    d["write_bytes"] = d.get("WRITE_SIZE", 0) * 1024
    d["hbm_bytes_per_launch"] = d["read_bytes"] + d["write_bytes"]
    summary["kernels"][k] = d
json.dump(summary, open(os.path.join(dst, f"{rnd}_pmc_summary.json"), "w"), indent=1)

# bench.py keys: "<mode>_border<b>_frames<n>"; the profiled run is the default bench config
main = [k for k in per_kernel if "k_reproject_pack" in k]
if main:
    # the default config launches the VEC stereo kernel 5+1+... times; variants use the same kernel with border 0,
    # so take the FIRST dispatches (main run) from the raw CSV instead of the mean
    f = one(f"{tag}_fetch/*/*_counter_collection.csv"); w = one(f"{tag}_write/*/*_counter_collection.csv")
    def first_n(path, ctr, n=6):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "k_reproject_pack" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
        return vals[:n]
    fv, wv = first_n(f, "FETCH_SIZE"), first_n(w, "WRITE_SIZE")
    rb = sum(fv) / len(fv) * 1024 * round(fetch_factor, 3); wb = sum(wv) / len(wv) * 1024
    traffic["parity_border40_frames16"] = {
        "hbm_bytes_per_launch": rb + wb, "read_bytes": rb, "write_bytes": wb,
        "build": build_id(),
        "source": f"profiles/{rnd}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                  f"FETCH_SIZE x{round(fetch_factor,3)} per calibration)"}
    cf = one(f"{tag}_cfetch/*/*_counter_collection.csv"); cw = one(f"{tag}_cwrite/*/*_counter_collection.csv")
    if cf and cw:
        def mean_of(path, ctr, kern):
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kern in r["Kernel_Name"] and r["Counter_Name"] == ctr]
            return sum(vals) / len(vals) if vals else 0.0
        kerns = ("k_compact_count", "k_compact_scan", "k_compact_scatter", "k_compact_onepass")
        rb = sum(mean_of(cf, "FETCH_SIZE", k) for k in kerns) * 1024 * round(fetch_factor, 3)
        wb = sum(mean_of(cw, "WRITE_SIZE", k) for k in kerns) * 1024
        traffic["compact_border40_frames16"] = {
            "hbm_bytes_per_launch": rb + wb, "read_bytes": rb, "write_bytes": wb, "build": build_id(),
            "source": f"profiles/{rnd}_pmc_summary.json (all compaction kernels of one step: the single-pass kernel "
                      f"for launches of >= 4 frames and >= 24576 tiles, else count + scan + scatter)"}
    json.dump(traffic, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
print(json.dumps(traffic, indent=1))
