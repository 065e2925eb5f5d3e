#!/bin/bash
# PARITY launch shape for the reference's REAL input types (cpp:60-61: 8-bit disparities times 1/8; 16-bit alike): the
# one-shot blocks of 256 x pxt pixels (pxt 1, 2; 4 with --small 1) against the tile-walking kernel (pxt 4, 8), per dtype.
for dt in u8 u16 f32; do
for geo in "1 752 480" "1 3840 2160" "64 752 480" "16 3840 2160"; do
  set -- $geo
  for b in 40; do
    echo "== $dt  $1 x $2x$3 border $b"
    python tools/ab.py --libs base --modes parity --dtype $dt --pxts 1,2,4,8 --bpcs 128 --borders $b --frames $1 --w $2 --h $3 --iters 20 2>&1 | grep -v amdgpu | sed 's/ novec=0 algo=1 oalign=16 ooff=0 form=0//'
    python tools/ab.py --libs base --modes parity --dtype $dt --pxts 4 --small 1 --bpcs 128 --borders $b --frames $1 --w $2 --h $3 --iters 20 2>&1 | grep -v amdgpu | sed 's/ novec=0 algo=1 oalign=16 ooff=0 form=0/ (small)/'
  done
done
done
