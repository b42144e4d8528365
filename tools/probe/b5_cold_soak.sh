#!/bin/bash
# Cold-launch soak of the one-launch fused backward: N fresh processes of tools/probe/b5_cold_soak.py at K = 8 and K = 10.
set -u
mkdir -p gpurun_out/b5_soak
N=${N:-50}
for k in 8 10; do
  log=gpurun_out/b5_soak/k$k.log
  : > $log
  for i in $(seq 1 $N); do
    timeout -k 10 120 python tools/probe/b5_cold_soak.py --k $k >> $log 2>&1 || echo "rc=$?" >> $log
  done
  echo "one-launch backward, K = $k: $(grep -c 'clean' $log) clean of $N; bad: $(grep -c BAD $log)"
done
