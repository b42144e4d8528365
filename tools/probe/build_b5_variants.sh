#!/bin/bash
# Variant builds of the one-launch backward (fc_rq_fused_backward512.hip) into tools/probe/build/:
#   tools/probe/build_b5_variants.sh name "<hipcc -D flags>" [name flags ...]
# e.g. abl4 "-DFC_B5_ABL=4" (ablation bits in fc_rq_fused_backward512.h).  Only the K = 8 / linear-tails instance is built.
# Run one with  python tools/bench_kernel.py --lib tools/probe/build/libb5_<name>.so --log2n 19 fused_bwd_wide
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$ROOT/flowconductor_amd/csrc"
make -s
OUT=$ROOT/tools/probe/build
mkdir -p $OUT
OTHERS=$(ls *.o | grep -v "^fc_rq_fused_backward512.o\$")
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DFC_B5_ONLY_K8 $flags -I. -c fc_rq_fused_backward512.hip -o $OUT/b5_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libb5_$name.so $OUT/b5_$name.o $OTHERS
  echo built $OUT/libb5_$name.so
done
