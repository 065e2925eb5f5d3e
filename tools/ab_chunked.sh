#!/bin/bash
# A/B of the big-batch COMPACT forms, interleaved in ONE process per workload: single pass (algo 2) against the chunked
# two-pass (algo 4) over chunk sizes (the tuning alternatives repeat for algo 2, which ignores them: its spread).
# usage: tools/ab_chunked.sh [libs] [tunes]
LIBS=${1:-base}
T=${2:-"chunk_mb=96;chunk_mb=64;chunk_mb=32;chunk_mb=128;chunk_mb=96,chunk_first_frames=4"}
for args in "--holes 0 --idx 0" "--holes 0.3 --idx 0" "--holes 0.3 --idx 1" "--holes 0.3 --blocky 1 --idx 1" "--holes 0.3 --idx 1 --frames 32 --w 1920 --h 1080"; do
  echo "== $args"
  python tools/ab.py --libs $LIBS --modes compact --algos 2,4 --pxts 8 --opbpc 0 --tunes "$T" $args 2>&1 | grep -v amdgpu.ids
done
