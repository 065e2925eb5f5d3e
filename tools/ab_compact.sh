#!/bin/bash
# A/B of single-pass compaction builds (interleaved, one process): all-valid, 30 % holes + indices, 32 x 1080p.
# usage: tools/ab_compact.sh base,r1 [opbpc]
LIBS=${1:-base,r1}; BPC=${2:-4}
python tools/ab.py --libs $LIBS --modes compact --algos 2 --pxts 8 --holes 0 --idx 0 --opbpc $BPC 2>&1 | grep -v amdgpu.ids
python tools/ab.py --libs $LIBS --modes compact --algos 2 --pxts 8 --holes 0.3 --idx 1 --opbpc $BPC 2>&1 | grep -v amdgpu.ids
python tools/ab.py --libs $LIBS --modes compact --algos 2 --pxts 8 --holes 0.3 --idx 1 --frames 32 --w 1920 --h 1080 --opbpc $BPC 2>&1 | grep -v amdgpu.ids
