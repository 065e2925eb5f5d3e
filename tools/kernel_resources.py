#!/usr/bin/env python3
"""Table of the compiler's register / scratch / LDS report of one source (`make -C disparity_to_point_cloud_amd/csrc asm
SRC=d2pc_callback` writes d2pc_callback.s.resources):  python tools/kernel_resources.py <file.s.resources> [regex]"""
import re, subprocess, sys

txt = open(sys.argv[1]).read()
filt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: Function Name: ", txt)[1:]
names = [b.split()[0] for b in blocks]
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
print("%-84s %5s %5s %7s %4s %6s %6s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "sgprSp", "vgprSp", "LDS"))
for b, n in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    n = re.sub(r"\(.*", "", n).replace("void d2pc::", "")
    if filt and not re.search(filt, n):
        continue
    print("%-84s %5d %5d %7d %4d %6d %6d %6d" % (n[:84], g("VGPRs"), g("TotalSGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                                                  g(r"Occupancy \[waves/SIMD\]"), g("SGPRs Spill"), g("VGPRs Spill"),
                                                  g(r"LDS Size \[bytes/block\]")))
