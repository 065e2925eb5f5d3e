#!/bin/bash
# (round 6) Do the single pass's 16 WRITE FRONTS collide in the memory system?  They start exactly
# out_frame_stride x 16 B apart (bench: 125,132,800 B = 8 KiB x 15,275; here the full frame, 132,710,400 B = 8 KiB x 16,200) and
# advance in lock-step (a block serves frame blockIdx % n_frames).  Interleaved on one device, one set of buffers:
#   oextra  the output frame stride padded by an ODD number of 4-KiB pages (256 points each): fronts no longer a multiple of 8 KiB apart
#   split   the 16 frames as 2 / 4 sub-launches of 8 / 4 frames back to back: only 8 / 4 fronts live at a time
#           (compact_algo 2 forced: a 4-frame launch is below the default routing's single-pass threshold)
# usage: tools/ab_fronts.sh  ->  profiles/r06_ab_fronts.txt
run() { python tools/ab.py --libs base --modes compact --algos 2 --pxts 8 --opbpc 0 --rounds 9 --iters 20 "$@" 2>&1 | grep -v amdgpu.ids | sed 's/ b=40 pxt= 8 bpc=128 novec=0 algo=2 oalign=16 ooff=0 form=0//'; }
for H in 0.3 0; do
  for I in 0 1; do
    echo "== 16 x 4K, holes $H, indices $I: stride padding"; run --holes $H --idx $I --oextras 0,256,768,1792,7936,65792
    echo "== 16 x 4K, holes $H, indices $I: sub-launches"; run --holes $H --idx $I --splits 1,2,4
  done
done
echo "== PARITY 16 x 4K (one-shot blocks: the control): stride padding"
python tools/ab.py --libs base --modes parity --pxts 0 --rounds 9 --iters 20 --oextras 0,256,1792 2>&1 | grep -v amdgpu.ids
