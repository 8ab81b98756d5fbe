#!/bin/bash
# round 5, GPU call G: the sequence table on the final code, incl. the bf16 / bf16_colour trackers.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5g; mkdir -p $O
timeout 1200 python tools/synthetic_sequence.py > $O/sequence.json 2> $O/sequence_err.log; echo "seq rc=$?" >> $O/sequence_err.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5g/sequence.json'))
for k,v in d.items():
    if k.startswith('free_running'): print(k, json.dumps(v)[:300])
PY
tail -2 $O/sequence_err.log
