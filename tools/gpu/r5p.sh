#!/bin/bash
# round 5, GPU call P: the mini-SLAM table with the schedule "fine" on both engines.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5p; mkdir -p $O
timeout 1500 python tools/synthetic_sequence.py --slam --frames 50 --schedule fine --verbose > $O/slam.json 2> $O/slam_err.log; echo "rc=$?" >> $O/slam_err.log
grep "frame\|engine\|rc=" $O/slam_err.log | grep -v Warn | grep "mapped\|engine\|rc=" ; cat $O/slam.json
