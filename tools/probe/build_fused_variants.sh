#!/bin/bash
# Ablation builds of the symmetric fused kernel (fc_rq_fused2.hip, macro FC_ABL) into tools/probe/build/.
# Run a variant with  FLOWCON_HIP_LIB=tools/probe/build/libfc_abl<N>.so python tools/bench_kernel.py fused
set -e
cd "$(dirname "$0")/../../flowconductor_amd/csrc"
make -s
OUT=../../tools/probe/build
mkdir -p $OUT
OTHERS=$(ls *.o | grep -v '^fc_rq_fused2.o$')
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DFC_ABL=$v ${EXTRA:-} -c fc_rq_fused2.hip -o $OUT/fused2_abl$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libfc_abl$v.so $OUT/fused2_abl$v.o $OTHERS
  echo built $OUT/libfc_abl$v.so
done
