#!/bin/bash
# Cold-launch soak of the round-4 kernels: N fresh processes of tools/probe/r4_cold_soak.py (one seed each).
set -u
mkdir -p gpurun_out/r4_soak
N=${N:-40}
log=gpurun_out/r4_soak/soak.log
: > $log
for i in $(seq 1 $N); do
  timeout -k 10 120 python tools/probe/r4_cold_soak.py $i >> $log 2>&1 || echo "rc=$?" >> $log
done
echo "round-4 kernels: $(grep -c 'clean' $log) clean of $N; bad: $(grep -c BAD $log); non-zero exits: $(grep -c '^rc=' $log)"
grep BAD $log | head -5
