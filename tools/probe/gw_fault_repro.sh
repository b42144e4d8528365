#!/bin/bash
# Cold-launch repro of the round-2 gW fault on probe builds of the OLD fc_rq_fused_linear_backward (tools/probe/build/libfc_oldbwd_*.so:
# the current library with the header of commit 44d992e^): N fresh processes per variant, one log per variant under gpurun_out/gw_fault/.
set -u
mkdir -p gpurun_out/gw_fault
N=${N:-6}
for v in "$@"; do
  log=gpurun_out/gw_fault/$v.log
  : > $log
  for i in $(seq 1 $N); do
    echo "== run $i" >> $log
    timeout -k 10 120 python tools/probe/cold_launch_gw.py --lib tools/probe/build/libfc_oldbwd_$v.so >> $log 2>&1 || echo "rc=$?" >> $log
  done
  echo "$v: $(grep -c 'first launch: bad rows \[\]' $log) clean of $N"
done
