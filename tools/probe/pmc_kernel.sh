#!/bin/bash
# SQ counters of one kernel of tools/bench_kernel.py <kind> (kernel symbol substring <match>), a few counters per pass.
# Usage on the GPU box: bash tools/probe/pmc_kernel.sh <tag> <kind> <match>
set -u
TAG=${1:-x}
KIND=${2:-fused}
MATCH=${3:-rq_fused_linear}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_${KIND}_$TAG
mkdir -p $OUT
cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F32" \
           "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_kernel.py --log2n ${LOG2N:-20} $KIND > $OUT/p$i.log 2>&1
  echo "pass $i exit $?"
done
cd $R
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$MATCH" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
import hashlib
sha = hashlib.sha256(open("$R/flowconductor_amd/csrc/libflowcon_hip.so", "rb").read()).hexdigest()
with open("$OUT/summary.txt", "w") as o:
    o.write("library.sha256 %s  (flowconductor_amd/csrc/libflowcon_hip.so; bench.py compares it with the loaded library)\n" % sha)
    for k in sorted(acc):
        v = acc[k]
        line = "%-32s mean/launch %.4g  (launches %d)" % (k, sum(v) / len(v), len(v))
        print(line); o.write(line + "\n")
PY
