#!/bin/bash
# round 5, GPU call K: final state with the resident-weight bf16 quad kernels: full GPU suite, smoke, both bench commands, bf16 rocprof.
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
O=$R/gpurun_out/r5k; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$?" >> $O/bench_err.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_command.json 2>> $O/bench_err.log
cd /tmp && export TMPDIR=/tmp
B16="python $R/bench.py --precision bf16 --no-cpu-baseline --no-mapping --no-dropin --no-precision-modes --steps 200 --warmup 20"
(timeout 300 $B16) > $O/bf16_bench_line.json 2> /tmp/b16.err
rm -rf /tmp/k16; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k16 -- $B16 > /tmp/k16.log 2>&1
cp $(find /tmp/k16 -name "*kernel_stats.csv" | head -1) $O/bf16_bench_kernel_stats.csv
cd $R
grep -n "passed\|failed" $O/gpu_tests.log | tail -2; tail -2 $O/smoke.log; cut -c1-200 $O/bench_line.json; cut -c1-200 $O/bench_line_driver_command.json; cut -c1-200 $O/bf16_bench_line.json; head -9 $O/bf16_bench_kernel_stats.csv | cut -c1-150
