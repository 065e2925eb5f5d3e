#!/bin/bash
# plain against nt point stores of the ragged pieces (libd2pc.so against libd2pc_ntst.so = make variant NAME=ntst
# DEFS="-DD2PC_CHUNK_STORE_NT=1 -DD2PC_RESIDENT_STORE_NT=1"), interleaved
for args in "--frames 1 --holes 0.3 --idx 1 --algos 3" "--frames 1 --holes 0.3 --idx 0 --algos 3" "--frames 1 --holes 0 --idx 0 --algos 3" "--frames 2 --holes 0.3 --idx 1 --algos 3" "--frames 16 --holes 0.3 --idx 1 --algos 2,4" "--frames 16 --holes 0.3 --idx 0 --algos 2,4" "--frames 16 --holes 0 --idx 0 --algos 2,4"; do
  echo "== $args"
  python tools/ab.py --libs base,ntst --modes compact --pxts 8 --opbpc 0 --rounds 9 --iters 10 $args 2>&1 | grep -v amdgpu.ids
done
