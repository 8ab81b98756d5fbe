#!/bin/bash
# full verification of HEAD + mapping profile + interleaved A/B
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5C; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$?" >> $O/bench_err.log
timeout 600 python tools/ab_mapping.py 30 3 > $O/ab.txt 2> $O/ab.err
bash tools/profile_mapping.sh > $O/profile.log 2>&1
cp gpurun_out/prof/mapping_kernel_stats.csv gpurun_out/prof/mapping_pmc_per_kernel.csv gpurun_out/prof/mapping_pmc_meta.json $O/
grep -n "passed\|failed" $O/gpu_tests.log | tail -2; tail -2 $O/smoke.log; cut -c1-200 $O/bench_line.json; tail -1 $O/bench_err.log; head -6 $O/ab.txt | cut -c1-120
