#!/usr/bin/env python3
"""Summarise per-kernel resource usage from a hipcc -S device assembly file."""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)(?=\n  - \.agpr_count|\Z)", txt, re.S):
    pass
# the YAML metadata lists kernels with fields; parse loosely
blocks = txt.split("  - .agpr_count:")[1:]
rows = []
for b in blocks:
    def g(k):
        mm = re.search(r"\.%s:\s+(\S+)" % k, b)
        return mm.group(1) if mm else "?"
    name = g("name")
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    dem = re.sub(r"\(.*", "", dem).replace("void d2pc::", "")
    rows.append((dem, g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
print("%-48s %5s %5s %6s %6s %7s %5s" % ("kernel", "vgpr", "sgpr", "vspill", "sspill", "scratch", "lds"))
for r in sorted(rows):
    print("%-48s %5s %5s %6s %6s %7s %5s" % r)
