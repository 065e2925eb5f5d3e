#!/bin/bash
# Two-pass (algo 1) against single-pass (algo 2) compaction over launch sizes, 30 % iid holes + indices,
# interleaved in one process (tools/ab.py).  Decides the default rule in enqueue() (d2pc_capi_route.hip).
for spec in "4 1920 1080" "8 1920 1080" "16 1920 1080" "32 1920 1080" "64 752 480" "256 752 480" "2 3840 2160" "4 3840 2160" "8 3840 2160" "16 3840 2160"; do
  set -- $spec
  echo "== $1 x $2x$3"
  python tools/ab.py --libs base --modes compact --algos 1,2 --pxts 8 --holes 0.3 --idx 1 --frames $1 --w $2 --h $3 --opbpc 4 --rounds 6 2>&1 | grep -v amdgpu.ids
done
