#!/bin/bash
# The round's profile set (GPU box): rocprofv3 per-kernel stats of the headline run, and the HBM traffic
# counters in separate --pmc passes (kernel-trace only).  tools/collect_profiles.py rNN then condenses
# gpurun_out/prof_rN_* into profiles/.  Usage: tools/profile_round.sh 2
set -e
N=${1:-2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof_r${N}
# headline only (--no-variants): k_reproject_pack<F32>'s average is then the border-40 workload alone, and the
# bench line printed inside this run is the one the stats must agree with
rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_stats -o run -- python3 bench.py --steps 50 --warmup 5 --no-cpu --no-variants --no-extras > gpurun_out/bench_r${N}_under_rocprof.json 2> gpurun_out/bench_r${N}_under_rocprof.err
echo "stats (headline) done"
rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_stats_variants -o run -- python3 bench.py --steps 50 --warmup 5 --no-cpu --no-host-path > gpurun_out/bench_r${N}_under_rocprof_variants.json 2>> gpurun_out/bench_r${N}_under_rocprof.err
echo "stats (variants) done"
for c in FETCH_SIZE WRITE_SIZE; do
  d=$( [ $c = FETCH_SIZE ] && echo fetch || echo write )
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d ${P}_$d -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-variants --no-extras > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d ${P}_c$d -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-variants --no-extras --mode compact > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d ${P}_calib_$d -o run -- ./tools/membench calib > /dev/null 2>&1
  echo "pmc $c done"
done
