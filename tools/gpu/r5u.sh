#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5u; mkdir -p $O
timeout 900 python -m pytest tests/test_precision_gpu.py -q -rf -s > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -n "resident vs staged" $O/tests.log | cut -c1-400; grep -n "passed\|failed\|Error\|assert\|rc=" $O/tests.log | cut -c1-250 | tail -12
