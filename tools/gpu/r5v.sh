#!/bin/bash
# final verification of HEAD as the driver runs it
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5v; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$?" >> $O/bench_err.log
grep -n "passed\|failed" $O/gpu_tests.log | tail -2; tail -2 $O/smoke.log; cut -c1-200 $O/bench_line.json; tail -1 $O/bench_err.log
