#!/bin/bash
# round 5, GPU call O: one frame-5 mapping round from the same state on the fused and the composed engine, loss terms side by side.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5o; mkdir -p $O
timeout 900 python tools/diag_slam_mapping.py > $O/diag.log 2> $O/diag_err.log; echo "rc=$?" >> $O/diag_err.log
tail -3 $O/diag_err.log; cat $O/diag.log | cut -c1-260
