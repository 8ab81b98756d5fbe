#!/bin/bash
# round 5, GPU call L: why the driver's 20-step command reads ~4 % above the 200-step one: cache pre-warm length.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5l; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-mapping --no-dropin --no-precision-modes"
for spec in "20 5 100" "20 5 400" "20 5 1000" "200 20 100" "20 5 100" "20 5 1000" "60 5 100"; do set -- $spec
  (timeout 300 $B --steps $1 --warmup $2 --prewarm-steps $3 | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'steps': $1, 'warmup': $2, 'prewarm_steps': $3, 'ms': l['ms_per_step'], 'sampler_us': l['roofline']['avg_launch_us']}))") >> $O/prewarm.jsonl 2>> $O/err.log
done
cat $O/prewarm.jsonl
