#!/bin/bash
# round 5, GPU call A: full GPU suite (new oracle-bf16 / sequence / drop-in tests included), default bench line, bf16 sampler A/Bs,
# the 50-frame synthetic sequence.  Run ON THE GPU BOX via gpurun from the repo root.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -m gpu -q -rf > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$?" >> $O/bench_err.log
for spec in "fp32 64" "bf16 64" "bf16 32"; do set -- $spec
  NSA_SAMPLER_TILE=$2 timeout 300 python tools/ab_kernels.py --precision $1 >> $O/ab.jsonl 2>> $O/ab_err.log
done
NSA_LIB_TAG=b3 NSA_SAMPLER_TILE=32 timeout 300 python tools/ab_kernels.py --precision bf16 >> $O/ab.jsonl 2>> $O/ab_err.log
NSA_SAMPLER_TILE=32 timeout 300 python tools/ab_kernels.py --precision bf16 --rays 512 >> $O/ab.jsonl 2>> $O/ab_err.log
NSA_SAMPLER_TILE=64 timeout 300 python tools/ab_kernels.py --precision bf16 --rays 512 >> $O/ab.jsonl 2>> $O/ab_err.log
timeout 1200 python tools/synthetic_sequence.py > $O/sequence.json 2> $O/sequence_err.log; echo "seq rc=$?" >> $O/sequence_err.log
tail -3 $O/gpu_tests.log; cut -c1-300 $O/bench_line.json; cat $O/ab.jsonl | cut -c1-400; head -c 1500 $O/sequence.json
