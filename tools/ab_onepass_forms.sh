#!/bin/bash
# (round 5) the single pass in its kernel forms, interleaved in one process on one set of buffers:
#   onepass_form 1 = raw tiles in LDS, every pixel decided in both phases (rounds 2-4)
#   onepass_form 2 = survivors packed by the count phase, dense scatter, 4 worker waves (2,048-pixel tiles)
#   onepass_form 3 = the same with 8 worker waves (4,096-pixel tiles)
#   onepass_form 5 = form 2 with 4 runs per worker wave (4,096-pixel tiles, 2 blocks per CU); 6 = form 2 with deferred landing; 7 = both
# usage: tools/ab_onepass_forms.sh [libs] [forms]   (libs: exp = the experiment build, which holds every form)  ->  profiles/r05_ab_onepass_forms_*.txt
LIBS=${1:-exp}; T=${2:-"onepass_form=1;onepass_form=2;onepass_form=3"}
run() { python tools/ab.py --libs $LIBS --modes compact --algos 2 --pxts 8 --opbpc 0 --rounds 9 --iters 20 --tunes "$T" "$@" 2>&1 | grep -v amdgpu.ids | sed 's/ b=40 pxt= 8 bpc=128 novec=0 algo=2 oalign=16 ooff=0 form=0//'; }
echo "== 16 x 4K, all valid"; run --holes 0 --idx 0
echo "== 16 x 4K, all valid + indices"; run --holes 0 --idx 1
echo "== 16 x 4K, 30 % holes"; run --holes 0.3 --idx 0
echo "== 16 x 4K, 30 % holes + indices"; run --holes 0.3 --idx 1
echo "== 16 x 4K, 30 % holes in 64x64 blocks + indices"; run --holes 0.3 --blocky 1 --idx 1
echo "== 16 x 4K, 90 % holes + indices"; run --holes 0.9 --idx 1
echo "== 32 x 1080p, 30 % holes + indices"; run --holes 0.3 --idx 1 --frames 32 --w 1920 --h 1080
echo "== 32 x 1080p, all valid + indices"; run --holes 0 --idx 1 --frames 32 --w 1920 --h 1080
echo "== 16 x 4K u8, 30 % holes + indices"; run --holes 0.3 --idx 1 --dtype u8
