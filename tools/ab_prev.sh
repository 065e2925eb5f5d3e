#!/bin/bash
# this build against the previous one (cp libd2pc.so libd2pc_prev.so before rebuilding), interleaved: the COMPACT kernels
for args in "--frames 16 --holes 0.3 --idx 1 --algos 2" "--frames 16 --holes 0.3 --idx 0 --algos 2" "--frames 16 --holes 0 --idx 0 --algos 2" "--frames 32 --holes 0.3 --idx 1 --algos 2 --w 1920 --h 1080" "--frames 1 --holes 0.3 --idx 1 --algos 3" "--frames 1 --holes 0.3 --idx 1 --algos 3 --w 1920 --h 1080" "--frames 4 --holes 0.3 --idx 1 --algos 1"; do
  echo "== $args"
  python tools/ab.py --libs base,prev --modes compact --pxts 8 --opbpc 0 --rounds 9 --iters 10 $args 2>&1 | grep -v amdgpu.ids | sed 's/ b=40 pxt= 8 bpc=128 novec=0//;s/ oalign=16 ooff=0 form=0//'
done
