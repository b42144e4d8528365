#!/bin/bash
# the old role-1 kernel and the product library under inherited-state poisoning (tools/probe/poisoned_launch_gw.py)
set -u
mkdir -p gpurun_out/gw_fault
timeout -k 10 300 python tools/probe/poisoned_launch_gw.py --lib tools/probe/build/libfc_oldbwd_v0.so --scan > gpurun_out/gw_fault/poisoned_old.log 2>&1; echo "old rc=$?"
timeout -k 10 300 python tools/probe/poisoned_launch_gw.py --scan --both-roles > gpurun_out/gw_fault/poisoned_current.log 2>&1; echo "current rc=$?"
timeout -k 10 300 python tools/probe/poisoned_launch_gw.py --scan --both-roles --k 10 > gpurun_out/gw_fault/poisoned_current_k10.log 2>&1; echo "current k10 rc=$?"
tail -30 gpurun_out/gw_fault/poisoned_old.log
