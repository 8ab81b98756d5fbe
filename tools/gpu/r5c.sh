#!/bin/bash
# round 5, GPU call C: the full GPU suite, the driver's bench command and the default one, rocprofv3 kernel stats + PMC passes of the
# fp32 bench (tools/profile_round.sh), rocprofv3 kernel stats of the bf16 bench.
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
O=$R/gpurun_out/r5c; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -m gpu -q -rf -x > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_command.json 2> $O/bench_err.log; echo "bench rc=$?" >> $O/bench_err.log
timeout 600 python bench.py > $O/bench_line.json 2>> $O/bench_err.log; echo "bench rc=$?" >> $O/bench_err.log
timeout 1500 bash tools/profile_round.sh > $O/profile_round.log 2>&1
cp gpurun_out/prof/* $O/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
B16="python $R/bench.py --precision bf16 --no-cpu-baseline --no-mapping --no-dropin --no-precision-modes --steps 200 --warmup 20"
(timeout 300 $B16) > $O/bf16_bench_line.json 2> /tmp/b16.err
rm -rf /tmp/k16; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k16 -- $B16 > /tmp/k16.log 2>&1
cp $(find /tmp/k16 -name "*kernel_stats.csv" | head -1) $O/bf16_bench_kernel_stats.csv
cd $R
tail -3 $O/gpu_tests.log; cut -c1-260 $O/bench_line_driver_command.json; cut -c1-260 $O/bench_line.json; cut -c1-260 $O/bf16_bench_line.json; head -12 $O/bf16_bench_kernel_stats.csv | cut -c1-160; cat $O/hbm_traffic.json
