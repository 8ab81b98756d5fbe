#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5r; mkdir -p $O
timeout 900 python -m pytest tests/test_sequence_gpu.py -q -rf -s > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -n "mini-SLAM\|ATE\|passed\|failed\|assert\|rc=" $O/tests.log | cut -c1-250
