#!/bin/bash
# EXPERIMENT: the register-resident one-launch form over MORE blocks than are resident at once (16 x 4K), against the
# single pass; interleaved.  (spin_timeout_ms=300: a dispatch-order violation shows as time-outs, not as a hang)
T="spin_timeout_ms=300;resident_pxt=32,resident_unbounded=1,spin_timeout_ms=300;resident_pxt=64,resident_unbounded=1,spin_timeout_ms=300"
for args in "--frames 16 --holes 0.3 --idx 1" "--frames 16 --holes 0.3 --idx 0" "--frames 16 --holes 0 --idx 0" "--frames 16 --holes 0.3 --blocky 1 --idx 1" "--frames 32 --holes 0.3 --idx 1 --w 1920 --h 1080"; do
  echo "== $args"
  timeout -k 10 240 python tools/ab.py --modes compact --algos 2,3 --pxts 8 --opbpc 0 --rounds 7 --iters 10 --tunes "$T" $args 2>&1 | grep -v amdgpu.ids
done
