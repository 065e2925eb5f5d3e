#!/usr/bin/env python3
"""Is the launch-to-launch spread of k_callback_bs clock or something else?  Reads one rocprofv3 --pmc GRBM_GUI_ACTIVE
--kernel-trace pass of tools/probe.py --workload callback_u8 (gpurun_out/pmc_cbvar/) and prints, per dispatch, the duration
and the effective clock GRBM_GUI_ACTIVE / 8 / duration (the counter is summed over the 8 XCDs: MI355X_MICROARCH.md, DVFS).
Usage: callback_variance.py <dir> [kernel-substring]"""
import csv, glob, os, sys
import numpy as np

d = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_callback_bs"
dur, act, names = {}, {}, {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if want in r["Kernel_Name"]:
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if want in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            act[int(r["Dispatch_Id"])] = float(r["Counter_Value"])
ids = sorted(set(dur) & set(act))[2:]  # the first two dispatches warm up
us = np.array([dur[i] for i in ids])
ghz = np.array([act[i] / 8.0 / (dur[i] * 1e3) for i in ids])
cyc = np.array([act[i] / 8.0 for i in ids])
print(f"# {want}: {len(ids)} dispatches; duration us min/median/max {us.min():.1f}/{np.median(us):.1f}/{us.max():.1f}"
      f"  effective clock GHz min/median/max {ghz.min():.3f}/{np.median(ghz):.3f}/{ghz.max():.3f}")
print(f"# busy cycles per XCD (GRBM_GUI_ACTIVE / 8) min/median/max {cyc.min():.0f}/{np.median(cyc):.0f}/{cyc.max():.0f}"
      f"  -> spread of duration {us.max() / us.min() - 1:+.1%}, of cycles {cyc.max() / cyc.min() - 1:+.1%}, of clock {ghz.max() / ghz.min() - 1:+.1%}")
print(f"# correlation duration ~ cycles {np.corrcoef(us, cyc)[0, 1]:+.2f}, duration ~ 1/clock {np.corrcoef(us, 1 / ghz)[0, 1]:+.2f}")
o = np.argsort(us)
for tag, sel in (("fastest", o[:3]), ("slowest", o[-3:])):
    for k in sel:
        print(f"{tag}: dispatch {ids[k]:4d}  {us[k]:7.1f} us  {cyc[k]:9.0f} cycles  {ghz[k]:.3f} GHz")
