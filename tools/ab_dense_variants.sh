#!/bin/bash
# (round 5) the dense single pass (onepass_form 2) in its build variants, interleaved:
#   base = two barriers per iteration, next tile landed in front of the scatter
#   ob   = ONE barrier per iteration (-DD2PC_DENSE_ONE_BARRIER=1)     ll = next tile landed BEHIND the scatter (-DD2PC_DENSE_LAND_LATE=1)
# usage: tools/ab_dense_variants.sh [libs] [blocks per CU]
LIBS=${1:-base,ob,ll,obll}; BPC=${2:-0}
run() { python tools/ab.py --libs $LIBS --modes compact --algos 2 --pxts 8 --opbpc $BPC --rounds 9 --iters 20 --tunes "onepass_form=2" "$@" 2>&1 | grep -v amdgpu.ids | sed 's/ b=40 pxt= 8 bpc=128 novec=0 algo=2 oalign=16 ooff=0 form=0//'; }
echo "== 16 x 4K, all valid (bpc $BPC)"; run --holes 0 --idx 0
echo "== 16 x 4K, 30 % holes (bpc $BPC)"; run --holes 0.3 --idx 0
echo "== 16 x 4K, 30 % holes + indices (bpc $BPC)"; run --holes 0.3 --idx 1
echo "== 16 x 4K, 90 % holes + indices (bpc $BPC)"; run --holes 0.9 --idx 1
echo "== 32 x 1080p, 30 % holes + indices (bpc $BPC)"; run --holes 0.3 --idx 1 --frames 32 --w 1920 --h 1080
