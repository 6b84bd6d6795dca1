#!/bin/bash
# Round-3 profile collection on the GPU box (run from the repository root through gpurun):
#   1. rocprofv3 --kernel-trace --stats over the default bench command (per-kernel durations);
#   2. one --pmc pass per counter over the same command (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES), never
#      combined with a trace domain; FETCH / WRITE once more with the per-layer residual kernels (ASW_NO_RESSTACK=1)
#      for the fused-pair comparison.
# The program itself follows `--` (no env / shell hop under the profiler).  Outputs: gpurun_out/r3prof/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r3prof
mkdir -p $OUT
BENCH="python3 $R/bench.py --no-e2e --no-extras --cpu-sample 0 --steps 5 --warmup 2"
BENCHP="python3 $R/bench.py --no-e2e --no-extras --cpu-sample 0 --steps 3 --warmup 1 --no-profile"
cd $R
echo "== kernel stats"; rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 1
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  echo "== pmc $c"; rocprofv3 --pmc $c -d $OUT/pmc_$c -o pmc -- $BENCHP > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err || exit 1
done
export ASW_NO_RESSTACK=1
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c (per-layer residual kernels)"; rocprofv3 --pmc $c -d $OUT/pmc_old_$c -o pmc -- $BENCHP > $OUT/pmc_old_$c.json 2> $OUT/pmc_old_$c.err || exit 1
done
unset ASW_NO_RESSTACK
find $OUT -name "*.csv" | head -30
