#!/bin/bash
# round 5, GPU call J: resident-weight persistent forms of the 32-point backward launches (bf16 build): tests, then A/B.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5j; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_precision_oracle_gpu.py tests/test_precision_gpu.py tests/test_tiling_gpu.py tests/test_track_fold_gpu.py -q -rf > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
for rep in 1 2; do for c in 1 0; do
  NSA_BF16_RESIDENT_CC=$c timeout 300 python tools/ab_kernels.py --precision bf16 --steps 100 >> $O/ab.jsonl 2>> $O/ab_err.log
done; done
for c in 1 0; do
  NSA_BF16_RESIDENT_CC=$c timeout 300 python tools/ab_kernels.py --precision bf16_colour --samples 192 --steps 100 >> $O/ab.jsonl 2>> $O/ab_err.log
done
grep -n "passed\|failed" $O/tests.log | tail -3; cut -c1-420 $O/ab.jsonl
