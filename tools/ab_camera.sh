#!/bin/bash
# (round 6) Camera-shaped COMPACT launches on DISTINCT frames: every launch takes the next frame(s) of a ring >= 512 MB, so its input comes
# from HBM, not from the Infinity Cache (the reference's operating mode: one new frame per callback, hpp:77-78).  Round 4 tuned the resident
# blocks' ramped start on ONE repeated (cached) frame; here the same sweep on uncached input, cached beside it.  Experiment build (the key
# "resident_stagger_pct" lives there).  usage: tools/ab_camera.sh  ->  profiles/r06_ab_camera.txt
T="resident_stagger_pct=0;resident_stagger_pct=25;resident_stagger_pct=50;resident_stagger_pct=100;resident_stagger_pct=150;resident_stagger_pct=250"
run() { python tools/ab.py --libs exp --modes compact --algos 3 --pxts 8 --rounds 9 --iters 32 "$@" 2>&1 | grep -v amdgpu.ids | sed 's/ b=40 pxt= 8 bpc=128 novec=0 algo=3 oalign=16 ooff=0 form=0//'; }
echo "== ONE 4K frame, 30 % holes + indices, ring of 16 distinct frames: ramped start"; run --frames 1 --ring 16 --holes 0.3 --idx 1 --tunes "$T"
echo "== the same frame again and again (cached input)"; run --frames 1 --ring 1 --holes 0.3 --idx 1 --tunes "$T"
echo "== ONE 4K frame, all valid, no indices, ring of 16"; run --frames 1 --ring 16 --holes 0 --idx 0 --tunes "$T"
echo "== ONE 4K frame, ring of 16: 32 against 64 pixels per thread"; run --frames 1 --ring 16 --holes 0.3 --idx 1 --tunes "resident_pxt=32;resident_pxt=64"
echo "== TWO 4K frames per call, ring of 8 pairs"; run --frames 2 --ring 8 --holes 0.3 --idx 1 --tunes "resident_stagger_pct=-1;resident_stagger_pct=0;resident_stagger_pct=50;resident_stagger_pct=100"
echo "== ONE 4K frame: two-pass (algo 1) and single pass (algo 2) on the ring, for reference"
python tools/ab.py --libs exp --modes compact --algos 1,2 --pxts 8 --opbpc 0 --rounds 7 --iters 32 --frames 1 --ring 16 --holes 0.3 --idx 1 2>&1 | grep -v amdgpu.ids
echo "== ONE 1080p frame, ring of 64 (k_compact_resident, ordinary tiles)"; run --frames 1 --ring 64 --holes 0.3 --idx 1 --w 1920 --h 1080
echo "== the same 1080p frame again and again"; run --frames 1 --ring 1 --holes 0.3 --idx 1 --w 1920 --h 1080
echo "== PARITY, one 4K frame, ring of 16 / cached (the bare stream for scale)"
python tools/ab.py --libs base --modes parity --pxts 0 --rounds 7 --iters 32 --frames 1 --ring 16 2>&1 | grep -v amdgpu.ids
python tools/ab.py --libs base --modes parity --pxts 0 --rounds 7 --iters 32 --frames 1 --ring 1 2>&1 | grep -v amdgpu.ids
