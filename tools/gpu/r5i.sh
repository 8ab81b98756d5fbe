#!/bin/bash
# round 5, GPU call I: resident-weight persistent quad kernels of the bf16 build: bf16 tests, then A/B resident vs staged.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5i; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_precision_oracle_gpu.py tests/test_precision_gpu.py tests/test_tiling_gpu.py tests/test_fused_gpu.py -q -rf > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
NSA_BF16_RESIDENT=0 timeout 600 python -m pytest tests/test_precision_oracle_gpu.py tests/test_precision_gpu.py -q -rf > $O/tests_staged.log 2>&1; echo "pytest rc=$?" >> $O/tests_staged.log
for rep in 1 2; do for r in 1 0; do
  NSA_BF16_RESIDENT=$r timeout 300 python tools/ab_kernels.py --precision bf16 --steps 100 >> $O/ab.jsonl 2>> $O/ab_err.log
done; done
for r in 1 0; do
  NSA_BF16_RESIDENT=$r timeout 300 python tools/ab_kernels.py --precision bf16 --rays 512 --steps 100 >> $O/ab.jsonl 2>> $O/ab_err.log
done
grep -n "passed\|failed" $O/tests.log $O/tests_staged.log | tail -4; cut -c1-420 $O/ab.jsonl
