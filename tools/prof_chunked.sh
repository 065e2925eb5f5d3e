#!/bin/bash
# Per-dispatch durations of the chunked two-pass (rocprofv3 kernel trace): usage tools/prof_chunked.sh <tag> [probe args...]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_chunk_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 tools/probe.py --launches 3 "$@" > $OUT.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/run_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-40:]:
    name = r["Kernel_Name"][:60]
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:10.1f} us  dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us  grid {r.get("Grid_Size", r.get("Grid_Size_X","?")):>10}  vgpr {r.get("VGPR_Count","?")} {name}')
PY
tail -2 $OUT.log
