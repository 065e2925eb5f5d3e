#!/bin/bash
# One / two 4K frames in COMPACT mode, interleaved in one process: two-pass (1) against the one-launch resident forms (3).
for args in "--frames 1 --holes 0.3 --idx 1" "--frames 1 --holes 0 --idx 0" "--frames 2 --holes 0.3 --idx 1" "--frames 1 --holes 0.3 --idx 1 --w 1920 --h 1080" "--frames 4 --holes 0.3 --idx 1 --w 1920 --h 1080"; do
  echo "== $args"
  python tools/ab.py --modes compact --algos 1,3 --pxts 8 --rounds 9 --iters 20 --tunes "resident_pxt=0;resident_pxt=32;resident_pxt=64" $args 2>&1 | grep -v amdgpu.ids
  python tools/ab.py --modes parity --pxts 2 --rounds 9 --iters 20 $args 2>&1 | grep -v amdgpu.ids
done
