#!/bin/bash
# round 5, GPU call Q: the mini-SLAM table over several seeds (fused x 5, composed x 2).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5q; mkdir -p $O
timeout 2400 python tools/synthetic_sequence.py --slam --frames 50 --schedule fine --engines "fused:11+12+13+14+15,composed:12+13" > $O/slam.json 2> $O/slam_err.log; echo "rc=$?" >> $O/slam_err.log
tail -2 $O/slam_err.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/r5q/slam.json'))
for k in ('slam_fused','slam_composed'):
    print(k, {x:d[k][x] for x in d[k] if x!='runs'}, [(r['seed'], round(r['ate_rmse_scene_units'],5), r['wall_s']) for r in d[k]['runs']])
print(d.get('slam_ate_ratio_fused_over_composed'))
PY
