#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5G; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_mapping_gpu.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -2 $O/tests.log
bash tools/profile_mapping.sh --stats-only > /dev/null 2>&1
cp gpurun_out/prof/mapping_kernel_stats.csv $O/
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5G/mapping_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows); nsa=sum(int(r['TotalDurationNs']) for r in rows if 'nsa::' in r['Name'])
print("total ms/it", tot/1e7, "nsa share", nsa/tot)
for r in rows:
    if 'emit' in r['Name']: print(f"{int(r['TotalDurationNs'])/1e4:8.1f} us/it {int(r['Calls'])/10:5.1f}  {r['Name'][:80]}")
PY
