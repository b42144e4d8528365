#!/bin/bash
# Profile `bench.py` on the GPU box: kernel-trace stats + separate PMC passes for HBM traffic.
# Usage (from the repo root, on the GPU box):  [STAGES="a b"] bash tools/profile_bench.sh <tag>
#   stage a: kernel trace + FETCH_SIZE / WRITE_SIZE passes of bench.py, traces of the configs and of the training step, summaries
#   stage b: SQ counters of the two hot kernels, then the un-profiled line with --strict-profiles (every profiles/<tag>_* file
#            must carry the sha256 of the library that line reports)
#   (tools/profile_configs.py <tag> [--sq]: the PMC traffic / SQ counters of every `configs` entry -- calls of their own; run them
#    BEFORE stage b, whose line then carries `configs.*.roofline.traffic` of the same library)
# Writes raw output under gpurun_out/prof_<tag>/ and the judged summaries under profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT $R/profiles
touch $OUT/.start          # (the manifest lists what this call writes after this moment)
STAGES=${STAGES:-"a b"}
ARGS="--steps 3 --warmup 3 --no-cpu-baseline --no-configs"   # 3 warm-up steps: the first two run 5-10 % slower under the profiler
if [[ " $STAGES " == *" a "* ]]; then
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace exit $?"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py $ARGS > $OUT/bench_$C.json 2> $OUT/bench_$C.err
  echo "$C exit $?"
done
# the secondary configurations (bench.py's `configs` block) and the training step: kernel-trace stats only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_configs -- python3 $R/tools/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err
echo "configs trace exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -- python3 $R/tools/probe/bench_train.py 19 --json > $OUT/train.json 2> $OUT/train.err
echo "train trace exit $?"
cd $R
python3 tools/summarize_profile.py $OUT $TAG
for what in configs train; do
  f=$(ls $OUT/trace_$what/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f profiles/${TAG}_${what}_kernel_stats.csv
done
cp $OUT/configs.jsonl profiles/${TAG}_configs_under_rocprof.jsonl 2>/dev/null
cp $OUT/train.json profiles/${TAG}_train_under_rocprof.json 2>/dev/null
fi
if [[ " $STAGES " == *" b "* ]]; then
cd $R
# SQ counters of the two hot kernels of the SAME library (the summaries carry its sha256)
bash tools/probe/pmc_kernel.sh $TAG fused rq_fused_linear_kernel3 > $OUT/pmc_fused.log 2>&1 && cp gpurun_out/pmc_fused_$TAG/summary.txt profiles/${TAG}_fused_sq_counters.txt
bash tools/probe/pmc_kernel.sh $TAG hidden resnet_hidden_kernel > $OUT/pmc_hidden.log 2>&1 && cp gpurun_out/pmc_hidden_$TAG/summary.txt profiles/${TAG}_hidden_sq_counters.txt
# the line of the profiled library, un-profiled, refusing any profile of another build: every profiles/${TAG}_* file must carry
# the sha256 that this line reports as `library.sha256`
python3 bench.py --steps 20 --warmup 5 --strict-profiles > profiles/${TAG}_bench_n1.json 2> $OUT/bench_final.err
echo "final strict bench exit $?"
fi
# the raw rocprofv3 tables cannot carry a field of their own: one manifest names every file of this tag with the library it
# was taken from (the JSON / text summaries also record it themselves)
cd $R
python3 - "$TAG" "$STAGES" "$OUT/.start" <<'PYEOF'
import glob, hashlib, json, os, sys, time
tag, stages = sys.argv[1], sys.argv[2]
lib = "flowconductor_amd/csrc/libflowcon_hip.so"
sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()
path = "profiles/%s_manifest.json" % tag
man = json.load(open(path)) if os.path.exists(path) else {"files": {}}
if man.get("library", {}).get("sha256") != sha:
    man = {"files": {}}            # another build: earlier entries no longer describe files of this library
man["library"] = {"path": lib, "sha256": sha}
fresh = os.path.getmtime(sys.argv[3])
for f in sorted(glob.glob("profiles/%s_*" % tag)):
    if f != path and os.path.getmtime(f) >= fresh:
        man["files"][os.path.basename(f)] = {"stage": stages, "bytes": os.path.getsize(f)}
json.dump(man, open(path, "w"), indent=1, sort_keys=True)
PYEOF
# gpurun only carries gpurun_out/ back: the judged files travel there too (copy them into profiles/ after the call)
# -- only what this call wrote (older files of the tag travelled here with the snapshot and may belong to another build)
mkdir -p $R/gpurun_out/profiles_$TAG && find $R/profiles -name "${TAG}_*" -newer $OUT/.start -exec cp {} $R/gpurun_out/profiles_$TAG/ \;
