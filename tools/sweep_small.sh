#!/bin/bash
# Tile size (pixels per thread) for SMALL launches: one or a few frames leave the chip with a single wave of
# blocks, so finer tiles may hide latency better.  Interleaved (tools/ab.py).
for spec in "1 752 480" "1 1920 1080" "1 3840 2160" "4 752 480" "2 1920 1080"; do
  set -- $spec
  echo "== $1 x $2x$3 parity";  python tools/ab.py --libs base --modes parity --pxts 4,8,16 --bpcs 128 --frames $1 --w $2 --h $3 --rounds 8 --iters 20 2>&1 | grep -v amdgpu.ids
  echo "== $1 x $2x$3 compact two-pass, 30% holes + idx"; python tools/ab.py --libs base --modes compact --algos 1 --pxts 4,8,16 --bpcs 128 --holes 0.3 --idx 1 --frames $1 --w $2 --h $3 --rounds 8 --iters 20 2>&1 | grep -v amdgpu.ids
done
