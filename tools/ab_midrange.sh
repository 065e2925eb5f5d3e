#!/bin/bash
# mid-size COMPACT launches (between "everything resident" and the single pass's big batches): two-pass (1), single pass (2), chunked two-pass (4)
for args in "--frames 3 --holes 0.3 --idx 1" "--frames 4 --holes 0.3 --idx 1" "--frames 4 --holes 0 --idx 0" "--frames 6 --holes 0.3 --idx 1" "--frames 8 --holes 0.3 --idx 1" "--frames 8 --holes 0.3 --idx 1 --w 1920 --h 1080" "--frames 16 --holes 0.3 --idx 1 --w 1920 --h 1080" "--frames 64 --holes 0.3 --idx 1 --w 752 --h 480"; do
  echo "== $args"
  python tools/ab.py --modes compact --algos 1,2,4 --pxts 8 --opbpc 0 --rounds 9 --iters 20 --tunes "chunk_mb=96;chunk_mb=32,chunk_first_frames=1" $args 2>&1 | grep -v amdgpu.ids | sed 's/ b=40 pxt= 8 bpc=128 novec=0//;s/ oalign=16 ooff=0 form=0//'
done
