#!/bin/bash
# The round's randomised soaks against the oracle on the current build (GPU box): every case compares the HIP path's bytes (points, indices,
# counts) with the CPU oracle's.  usage: tools/soak_round.sh <seed base>  ->  profiles/rNN_soaks.txt
S=${1:-1001}
run() { echo "== $*"; timeout -k 10 "$T" python "$@" 2>&1 | grep -v amdgpu.ids | tail -2; }
T=400 run tools/soak.py 1500 $((S+1))
T=120 run tools/stress_onepass.py 4000 $((S+2))
T=120 run tools/stress_onepass.py 1500 $((S+3)) contend
T=200 run tools/soak_callback.py 600 $((S+4))
T=200 run tools/soak_host.py 300 $((S+5))
T=120 run tools/soak_filters.py 800 $((S+6))
T=400 run tools/soak_lean.py 200 $((S+7))
