#!/usr/bin/env python3
"""profiles/valu_issue.json from a counters file of tools/collect_counters.py: per callback-body kernel the VALU wave-instructions
per launch (SQ_INSTS_VALU), the launch's cycles under the profiler (GRBM_GUI_ACTIVE / 8) and what follows from them -- stamped
with the BUILD the passes ran on (sha256/12 of csrc/*.hip|hpp, as bench.py's config.build), because bench.py only uses the
instruction count when that build is the one it has loaded.
Usage: make_valu_issue.py profiles/r06_callback_counters.json [out.json]"""
import hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_id():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "disparity_to_point_cloud_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:12]


def main():
    src = sys.argv[1]
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "valu_issue.json")
    d = json.load(open(src))
    out = {"how": "valu_issue_frac = SQ_INSTS_VALU x 2 cycles / (1,024 SIMDs x launch cycles): the share of the chip's vector-issue "
                  "slots a launch fills if every wave64 instruction took the 2 cycles of a SIMD-32 (fp64 and several integer forms take "
                  "4: a lower bound).  The instruction count is a property of the compiled kernel; bench.py combines it with the time and "
                  "the shader clock IT measures (d2pc_clock_probe_device) -- and only when `build` below is the build it has loaded.  The "
                  "figures under each kernel here are the profiler run's own (cycles = GRBM_GUI_ACTIVE / 8).",
           "build": build_id()}
    for label, w in d["workloads"].items():
        for k, e in w["kernels"].items():
            c, dv = e["counters"], e["derived"]
            if "SQ_INSTS_VALU" not in c or "GRBM_GUI_ACTIVE" not in c:
                continue
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            out[label] = {"kernel": k, "valu_wave_insts_per_launch": int(round(c["SQ_INSTS_VALU"])),
                          "gpu_cycles_per_launch": int(round(cyc)),
                          "valu_issue_frac": round(c["SQ_INSTS_VALU"] * 2.0 / 1024.0 / cyc, 4),
                          "simd_cycles_per_valu_wave_instruction": round(1024.0 * cyc / c["SQ_INSTS_VALU"], 3),
                          "lds_array_busy_frac": round(dv.get("lds_array_busy_frac(SQ_LDS_IDX_ACTIVE / 256 CUs / cycles)", 0.0), 4) or None,
                          "wait_any_frac_of_wave_cycles": round(dv.get("SQ_WAIT_ANY/SQ_WAVE_CYCLES", 0.0), 4) or None,
                          "effective_clock_ghz_under_pmc": round(dv.get("effective_clock_ghz_under_pmc", 0.0), 3) or None,
                          "source": f"{os.path.relpath(src, ROOT)} (tools/pmc_passes.sh, separate --pmc passes, 16 x 4K u8)"}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
