#!/bin/bash
# usage: tools/probe/run_fused_variants.sh 0 1 2 ...   (0 = product build)
for v in "$@"; do
  if [ "$v" = 0 ]; then lib=flowconductor_amd/csrc/libflowcon_hip.so; else lib=tools/probe/build/libfc_abl$v.so; fi
  echo -n "abl$v: "
  timeout -k 10 120 python tools/bench_kernel.py --lib $lib fused 2>&1 | grep median || exit 1
done
