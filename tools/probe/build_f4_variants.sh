#!/bin/bash
# Variant builds of the K-generic resident-weight fused kernel (fc_rq_fused4_k10.hip) into tools/probe/build/:
#   tools/probe/build_f4_variants.sh name "<hipcc -D flags>" [generator environment, e.g. FC_GEN_SHIFT=0.5] ...
# (triples of arguments).  Run one with  python tools/bench_kernel.py --lib tools/probe/build/libf4_<name>.so general_k10
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$ROOT/flowconductor_amd/csrc"
make -s
OUT=$ROOT/tools/probe/build
mkdir -p $OUT
OTHERS=$(ls *.o | grep -v "^fc_rq_fused4_k10.o\$")
while [ $# -ge 3 ]; do
  name=$1; flags=$2; genenv=$3; shift 3
  TMP=$OUT/f4src_$name
  mkdir -p $TMP
  cp fc_rq_fused4_k10.hip fc_rq_fused4_body.h $TMP/
  (cd "$ROOT" && env $genenv FC_GEN_OUT=$TMP/fc_rq_fused4_eval_k10.inc python tools/gen_fused_eval.py --bins 10 > /dev/null)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 $flags -I. -c $TMP/fc_rq_fused4_k10.hip -o $OUT/f4_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libf4_$name.so $OUT/f4_$name.o $OTHERS
  echo built $OUT/libf4_$name.so
done
