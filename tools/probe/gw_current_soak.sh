#!/bin/bash
# Cold-launch soak of the PRODUCT library's fc_rq_fused_linear_backward: N fresh processes of tools/probe/cold_launch_gw.py at K = 8
# and K = 10 (the check that found the round-2 fault on the old kernel at 3-20 % of cold launches).
set -u
mkdir -p gpurun_out/gw_fault
N=${N:-48}
for k in 8 10; do
  log=gpurun_out/gw_fault/current_k$k.log
  : > $log
  for i in $(seq 1 $N); do
    timeout -k 10 120 python tools/probe/cold_launch_gw.py --k $k >> $log 2>&1 || echo "rc=$?" >> $log
  done
  echo "current library, K = $k: $(grep -c 'first launch: bad rows \[\]' $log) clean of $N"
done
