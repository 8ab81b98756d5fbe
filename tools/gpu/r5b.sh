#!/bin/bash
# round 5, GPU call B: re-run of the recalibrated tests, bf16 single-piece staging A/B (product vs tag sap = all pieces), SLP bisect,
# the sequence table with per-iteration traces.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5b; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_precision_oracle_gpu.py tests/test_sequence_gpu.py tests/test_precision_gpu.py tests/test_tiling_gpu.py tests/test_fused_gpu.py tests/test_track_fold_gpu.py tests/test_mapping_gpu.py -q -rf > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
for tag in "" sap "" sap; do
  NSA_LIB_TAG=$tag timeout 300 python tools/ab_kernels.py --precision bf16 >> $O/ab.jsonl 2>> $O/ab_err.log
done
NSA_LIB_TAG= timeout 300 python tools/ab_kernels.py --precision bf16_colour --samples 192 >> $O/ab.jsonl 2>> $O/ab_err.log
NSA_LIB_TAG=sap timeout 300 python tools/ab_kernels.py --precision bf16_colour --samples 192 >> $O/ab.jsonl 2>> $O/ab_err.log
NSA_LIB_TAG= timeout 300 python tools/ab_kernels.py --precision fp32 >> $O/ab.jsonl 2>> $O/ab_err.log
timeout 900 bash tools/slp_bisect.sh run > $O/slp_bisect.txt 2>&1
timeout 900 python tools/synthetic_sequence.py > $O/sequence.json 2> $O/sequence_err.log; echo "seq rc=$?" >> $O/sequence_err.log
grep -n "passed\|failed\|FAILED" $O/tests.log | tail -8; cat $O/ab.jsonl | cut -c1-420; cat $O/slp_bisect.txt | head -40
