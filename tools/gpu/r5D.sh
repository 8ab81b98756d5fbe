#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5D; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_mapping_gpu.py tests/test_loss_gpu.py tests/test_model_gpu.py tests/test_dropin_gpu.py tests/test_configs_gpu.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -3 $O/tests.log
bash tools/profile_mapping.sh --stats-only > $O/profile.log 2>&1
cp gpurun_out/prof/mapping_kernel_stats.csv $O/
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5D/mapping_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows); nsa=sum(int(r['TotalDurationNs']) for r in rows if 'nsa::' in r['Name'])
print("total ms/it", tot/1e7, "nsa share", nsa/tot, "launches/it", sum(int(r['Calls']) for r in rows)/10)
PY
timeout 300 python bench.py --only-mapping 10 --no-cpu-baseline > $O/map.json 2> $O/map.err; cut -c1-120 $O/map.json
