#!/bin/bash
# round 5, GPU call D: the data-path colour backward without forward recomputation (ReLU masks + sigmoid outputs in the save area):
# full GPU suite, then A/B against the previous library (tag r5c), fp32 and bf16, alternating.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5d; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -m gpu -q -rf > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
for rep in 1 2; do for tag in "" r5c; do
  NSA_LIB_TAG=$tag timeout 300 python tools/ab_kernels.py --precision fp32 --steps 100 >> $O/ab.jsonl 2>> $O/ab_err.log
done; done
for tag in "" r5c; do
  NSA_LIB_TAG=$tag timeout 300 python tools/ab_kernels.py --precision bf16 --steps 100 >> $O/ab.jsonl 2>> $O/ab_err.log
done
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$?" >> $O/bench_err.log
grep -n "passed\|failed\|FAILED" $O/gpu_tests.log | tail -8; cat $O/ab.jsonl | cut -c1-420; cut -c1-200 $O/bench_line.json
