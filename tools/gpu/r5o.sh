#!/bin/bash
# round 5, GPU call O: one frame-5 mapping round from the same state on the fused and the composed engine, loss terms side by side.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5o; mkdir -p $O
timeout 900 python tools/diag_slam_mapping.py composed > $O/diag_c.log 2> $O/diag_err.log; echo "rc=$?" >> $O/diag_err.log
timeout 900 python tools/diag_slam_mapping.py fused > $O/diag.log 2>> $O/diag_err.log
tail -3 $O/diag_err.log; grep -n "round\|state after\|it   0\|it  25\|it  50\|it  70\|it  99" $O/diag_c.log $O/diag.log | cut -c1-330
