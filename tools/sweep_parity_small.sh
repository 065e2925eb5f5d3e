#!/bin/bash
# PARITY: the one-shot small-tile kernel (pxt 1, 2; 4 with --small 1) against the tile-walking one (pxt 4, 8) over launch sizes.
for geo in "1 752 480" "1 1920 1080" "1 3840 2160" "4 3840 2160" "16 1920 1080" "64 752 480" "16 3840 2160"; do
  set -- $geo
  for b in 40 0; do
    echo "== $1 x $2x$3 border $b"
    python tools/ab.py --libs base --modes parity --pxts 1,2,4,8 --bpcs 128 --borders $b --frames $1 --w $2 --h $3 --iters 20 2>&1 | grep -v amdgpu | sed 's/ novec=0 algo=1 oalign=16 ooff=0//'
    python tools/ab.py --libs base --modes parity --pxts 4 --small 1 --bpcs 128 --borders $b --frames $1 --w $2 --h $3 --iters 20 2>&1 | grep -v amdgpu | sed 's/ novec=0 algo=1 oalign=16 ooff=0/ (small)/'
  done
done
