# Profiles of the training path (tag r02d): SQ counters of the two roles of fc_rq_fused_linear_backward at N = 2^19 and the
# kernel-trace stats of tools/probe/bench_train.py 19.  Usage on the GPU box: bash tools/probe/profile_backward.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd $R
LOG2N=19 bash tools/probe/pmc_kernel.sh r02d fused_bwd "rq_fused_backward_kernel<8, true, 0>" > gpurun_out/pmc_role0.log 2>&1
cp gpurun_out/pmc_fused_bwd_r02d/summary.txt gpurun_out/r02d_bwd_role0_sq_counters.txt
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_fused_bwd_r02d/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "rq_fused_backward_kernel<8, true, 1>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/r02d_bwd_role1_sq_counters.txt", "w") as o:
    for k in sorted(acc):
        v = acc[k]
        o.write("%-32s mean/launch %.4g  (launches %d)\n" % (k, sum(v) / len(v), len(v)))
PY
grep -h Kernel_Name -m1 -A0 gpurun_out/pmc_fused_bwd_r02d/p1/*/*counter_collection.csv | head -1; cut -d, -f1-12 gpurun_out/pmc_fused_bwd_r02d/p1/*/*counter_collection.csv | grep -o 'rq_fused_backward_kernel[^(]*' | sort | uniq -c
rm -rf gpurun_out/pmc_fused_bwd_r02d/p*
OUT=$R/gpurun_out/prof_r02d
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -- python3 $R/tools/probe/bench_train.py 19 --json > $OUT/train.json 2> $OUT/train.err
echo "train trace exit $?"
cd $R
f=$(ls $OUT/trace_train/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f gpurun_out/r02d_train_kernel_stats.csv
cp $OUT/train.json gpurun_out/r02d_train_under_rocprof.json
rm -rf $OUT/trace_train
