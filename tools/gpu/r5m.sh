#!/bin/bash
# round 5, GPU call M: mini-SLAM (tracking + mapping) on the synthetic sequence -- small debug run, then the 50-frame table.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5m; mkdir -p $O
timeout 600 python tools/synthetic_sequence.py --slam --frames 7 --height 68 --width 120 --small-colour-grid --map-iters 20 --iters 30 --verbose > $O/slam_small.json 2> $O/slam_small_err.log; echo "rc=$?" >> $O/slam_small_err.log
tail -5 $O/slam_small_err.log; cat $O/slam_small.json | head -60
if grep -q "rc=0" $O/slam_small_err.log; then
  timeout 1500 python tools/synthetic_sequence.py --slam --frames 50 --verbose > $O/slam.json 2> $O/slam_err.log; echo "rc=$?" >> $O/slam_err.log
  tail -4 $O/slam_err.log; cat $O/slam.json
fi
