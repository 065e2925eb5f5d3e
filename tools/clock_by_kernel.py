#!/usr/bin/env python3
"""Per kernel of one rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace pass: median duration and effective core clock
(GRBM_GUI_ACTIVE / 8 XCDs / duration).  Usage: clock_by_kernel.py <dir>"""
import collections, csv, glob, os, sys
import numpy as np

d = sys.argv[1]
dur, act, name = {}, {}, {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        i = int(r["Dispatch_Id"])
        dur[i] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name[i] = r["Kernel_Name"]
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            act[int(r["Dispatch_Id"])] = float(r["Counter_Value"])
groups = collections.OrderedDict()
for i in sorted(set(dur) & set(act)):
    groups.setdefault((name[i], ), []).append(i)
# consecutive runs of one kernel with the same grid are one variant: split by gaps in dispatch ids of that name
print(f"{'kernel':60s} {'n':>5s} {'us med':>9s} {'GHz med':>8s} {'GHz min':>8s} {'GHz max':>8s}")
run, last = [], None
def flush(run):
    if len(run) < 8:
        return
    ids = run[3:]
    us = np.array([dur[i] for i in ids])
    ghz = np.array([act[i] / 8.0 / (dur[i] * 1e3) for i in ids])
    print(f"{name[ids[0]][:60]:60s} {len(ids):5d} {np.median(us):9.1f} {np.median(ghz):8.3f} {ghz.min():8.3f} {ghz.max():8.3f}")
for i in sorted(set(dur) & set(act)):
    if last is not None and (name[i] != name[last]):
        flush(run)
        run = []
    run.append(i)
    last = i
flush(run)
