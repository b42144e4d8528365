#!/bin/bash
# Profile `bench.py` on the GPU box: kernel-trace stats + separate PMC passes for HBM traffic.
# Usage (from the repo root, on the GPU box):  bash tools/profile_bench.sh <tag>
# Writes raw output under gpurun_out/prof_<tag>/ and the judged summaries under profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT $R/profiles
ARGS="--steps 3 --warmup 3 --no-cpu-baseline"   # 3 warm-up steps: the first two run 5-10 % slower under the profiler
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace exit $?"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py $ARGS > $OUT/bench_$C.json 2> $OUT/bench_$C.err
  echo "$C exit $?"
done
cd $R
python3 tools/summarize_profile.py $OUT $TAG
