#!/bin/bash
# round 5, GPU call E: A/Bs on the final colour kernels (feature stores, launch bound), bf16 quad kernels with 4-wave workgroups,
# then tests touching the colour path, the default bench and the profiles of the final code.
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
O=$R/gpurun_out/r5e; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do for tag in "" sf cf2; do
  NSA_LIB_TAG=$tag timeout 300 python tools/ab_kernels.py --precision fp32 --steps 100 >> $O/ab.jsonl 2>> $O/ab_err.log
done; done
for tag in "" q4 "" q4; do
  NSA_LIB_TAG=$tag timeout 300 python tools/ab_kernels.py --precision bf16 --steps 100 >> $O/ab.jsonl 2>> $O/ab_err.log
done
timeout 1200 python -m pytest tests -m gpu -q -rf -x > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$?" >> $O/bench_err.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_command.json 2>> $O/bench_err.log
timeout 1500 bash tools/profile_round.sh > $O/profile_round.log 2>&1
mkdir -p $O/prof; cp gpurun_out/prof/* $O/prof/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
B16="python $R/bench.py --precision bf16 --no-cpu-baseline --no-mapping --no-dropin --no-precision-modes --steps 200 --warmup 20"
(timeout 300 $B16) > $O/bf16_bench_line.json 2> /tmp/b16.err
rm -rf /tmp/k16; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k16 -- $B16 > /tmp/k16.log 2>&1
cp $(find /tmp/k16 -name "*kernel_stats.csv" | head -1) $O/bf16_bench_kernel_stats.csv
cd $R
cat $O/ab.jsonl | cut -c1-400; tail -3 $O/gpu_tests.log; cut -c1-220 $O/bench_line.json; cat $O/prof/hbm_traffic.json
