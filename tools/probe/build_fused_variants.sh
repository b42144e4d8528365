#!/bin/bash
# Ablation builds of a fused kernel (FILE=fc_rq_fused3 by default; macro FC_ABL) into tools/probe/build/.
# Run a variant with  python tools/bench_kernel.py --lib tools/probe/build/libfc_abl<N>.so fused
set -e
cd "$(dirname "$0")/../../flowconductor_amd/csrc"
make -s
OUT=../../tools/probe/build
mkdir -p $OUT
FILE=${FILE:-fc_rq_fused3}
OTHERS=$(ls *.o | grep -v "^$FILE.o\$")
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DFC_ABL=$v ${EXTRA:-} -c $FILE.hip -o $OUT/${FILE}_abl$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libfc_abl$v.so $OUT/${FILE}_abl$v.o $OTHERS
  echo built $OUT/libfc_abl$v.so
done
