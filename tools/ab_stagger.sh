#!/bin/bash
# The ramped start of k_compact_resident_lean: block t begins its loads t x (block bytes / read rate) late.  Scale 0 (all at once) .. 300 %.
# (round 6: the key lives in the experiment build only -- --libs exp)
T="resident_stagger_pct=0;resident_stagger_pct=50;resident_stagger_pct=100;resident_stagger_pct=150;resident_stagger_pct=200;resident_stagger_pct=300"
for args in "--frames 1 --holes 0.3 --idx 1" "--frames 1 --holes 0 --idx 0" "--frames 1 --holes 0.3 --blocky 1 --idx 1" "--frames 2 --holes 0.3 --idx 1" "--frames 1 --holes 0.3 --idx 1 --dtype u8" "--frames 1 --holes 0.3 --idx 1 --w 2560 --h 1440"; do
  echo "== $args"
  python tools/ab.py --libs exp --modes compact --algos 3 --pxts 8 --rounds 9 --iters 20 --tunes "$T" $args 2>&1 | grep -v amdgpu.ids
done
