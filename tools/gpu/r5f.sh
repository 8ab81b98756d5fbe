#!/bin/bash
# round 5, GPU call F: ray chunks on their own streams (2 / 4) against one chunk, fp32 and bf16 at 1024 and 512 rays.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5f; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --no-cpu-baseline --no-mapping --no-dropin --no-precision-modes --steps 200 --warmup 20"
for c in 1 2 4 1 2; do
  (timeout 300 $B --chunks $c | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'chunks': $c, 'ms': l['ms_per_step'], 'ray_chunks': l['config']['ray_chunks']}))") >> $O/chunks.jsonl 2>> $O/err.log
done
for c in 1 2; do
  (timeout 300 $B --chunks $c --rays 512 | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'rays': 512, 'chunks': $c, 'ms': l['ms_per_step']}))") >> $O/chunks.jsonl 2>> $O/err.log
  (timeout 300 $B --chunks $c --precision bf16 | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'precision': 'bf16', 'chunks': $c, 'ms': l['ms_per_step']}))") >> $O/chunks.jsonl 2>> $O/err.log
done
cat $O/chunks.jsonl; tail -3 $O/err.log
