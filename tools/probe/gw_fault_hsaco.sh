#!/bin/bash
# The old role-1 kernel from externally assembled code objects (tools/probe/gw_asm_variants.py: the compiler's own assembly with
# one class of waits strengthened), N cold processes per variant through the shim library (FC_PROBE_HSACO).
set -u
mkdir -p gpurun_out/gw_fault
N=${N:-16}
for v in "$@"; do
  log=gpurun_out/gw_fault/hsaco_$v.log
  : > $log
  for i in $(seq 1 $N); do
    echo "== run $i" >> $log
    FC_PROBE_HSACO=tools/probe/build/gw_$v.hsaco timeout -k 10 120 python tools/probe/cold_launch_gw.py --lib tools/probe/build/libfc_oldbwd_shim.so >> $log 2>&1 || echo "rc=$?" >> $log
  done
  echo "$v: $(grep -c 'first launch: bad rows \[\]' $log) clean of $N"
done
