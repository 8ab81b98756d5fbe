#!/bin/bash
# round 5, GPU call N: where the composed engine loses the trajectory in the mini-SLAM: per-frame errors, 8 frames, both engines.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5n; mkdir -p $O
timeout 900 python tools/synthetic_sequence.py --slam --frames 8 --verbose > $O/slam8.json 2> $O/slam8_err.log; echo "rc=$?" >> $O/slam8_err.log
grep -v Warn $O/slam8_err.log | tail -40
