#!/bin/bash
# Rebuilds the probe libraries of DESIGN.md section 4d (the round-2 gW fault): the CURRENT csrc with the fc_rq_fused_backward.h of
# commit 44d992e^ (the last one with the fault), in a scratch directory; results in tools/probe/build/ (not tracked, travels with
# gpurun).  Variants:  v0 (as it was + FC_BWD_PARTIALS), dpp (FC_DPP_BUILTIN), poison (LDS poisoned), drain, forcezero
# (-mllvm -amdgpu-waitcnt-forcezero), mfmapad (-mllvm -amdgpu-mfma-padding-ratio=100), shim (role 1 / K = 8 loadable from an
# external code object: tools/probe/gw_asm_variants.py + gw_fault_hsaco.sh).  The PARTIALS / POISON / DRAIN / shim hunks are the
# ones of commit 44d992e and of this round's scratch tree; see the section for what each run showed.
#   bash tools/probe/build_old_bwd_variants.sh /tmp/hybrid
set -eu
W=${1:-/tmp/hybrid}
R=$(cd "$(dirname "$0")/../.." && pwd)
rm -rf "$W" && mkdir -p "$W/flowconductor_amd" "$R/tools/probe/build"
cp -r "$R/flowconductor_amd/csrc" "$W/flowconductor_amd/" && cp -r "$R/include" "$W/"
cd "$W/flowconductor_amd/csrc" && rm -f *.o *.so
# (round 4: the product library no longer carries the two-launch roles; their instantiation files and C entry come from history)
git -C "$R" show 6d125f7:flowconductor_amd/csrc/fc_rq_fused_backward_tails.hip > fc_rq_fused_backward_tails.hip
git -C "$R" show 6d125f7:flowconductor_amd/csrc/fc_rq_fused_backward_box.hip > fc_rq_fused_backward_box.hip
git -C "$R" show 6d125f7:flowconductor_amd/csrc/fc_rq_fused_backward.hip > fc_rq_fused_backward.hip
rm -f fc_rq_fused_backward512.hip fc_rq_fused_backward512.h
git -C "$R" show 44d992e^:flowconductor_amd/csrc/fc_rq_fused_backward.h > fc_rq_fused_backward.h
python3 "$R/tools/probe/patch_old_bwd.py" fc_rq_fused_backward.h
build() { name=$1; shift; rm -f fc_rq_fused_backward_tails.o fc_rq_fused_backward_box.o fc_rq_fused_backward.o libflowcon_hip.so
  make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -DFC_BWD_PARTIALS $*" > /dev/null
  cp libflowcon_hip.so "$R/tools/probe/build/libfc_oldbwd_$name.so"; echo "built $name"; }
build v0
build dpp -DFC_DPP_BUILTIN
build poison -DFC_BWD_POISON=0x7fc07fc0u
build drain -DFC_BWD_DRAIN
build forcezero -mllvm -amdgpu-waitcnt-forcezero
build mfmapad -mllvm -amdgpu-mfma-padding-ratio=100
build shim -DFC_PROBE_HSACO_SHIM
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DFC_BWD_PARTIALS -S --cuda-device-only -o /tmp/probe_tails.s fc_rq_fused_backward_tails.hip
python3 "$R/tools/probe/gw_asm_variants.py"
