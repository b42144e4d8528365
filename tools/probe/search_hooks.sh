#!/bin/bash
# Builds probe variants of the fused kernel with shifted MFMA hook positions (FC_GEN_SHIFT) into
# tools/probe/build/libfc_shift<k>.so; the committed .inc is restored afterwards.
set -e
cd "$(dirname "$0")/../.."
for k in "$@"; do
  FC_GEN_SHIFT=$k python tools/gen_fused_eval.py > /dev/null
  tools/probe/build_fused_variants.sh 16 > /dev/null
  mv tools/probe/build/libfc_abl16.so tools/probe/build/libfc_shift$k.so
  echo built shift $k
done
python tools/gen_fused_eval.py > /dev/null
git diff --stat -- flowconductor_amd/csrc/fc_rq_fused3_eval.inc | tail -1
