#!/bin/bash
# emission rows: non-temporal (product) vs plain (tagged build) -- kernel stats of the mapping iteration, twice each, interleaved
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5F; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_mapping_gpu.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -2 $O/tests.log
for r in 1 2; do
  for t in emitplain ""; do
    NSA_LIB_TAG=$t bash tools/profile_mapping.sh --stats-only > /dev/null 2>&1
    cp gpurun_out/prof/mapping_kernel_stats.csv $O/stats_${t:-nt}_$r.csv
  done
done
python - <<'PY'
import csv, glob
for f in sorted(glob.glob('gpurun_out/r5F/stats_*.csv')):
    rows=list(csv.DictReader(open(f)))
    tot=sum(int(r['TotalDurationNs']) for r in rows)
    pick={k: sum(int(r['TotalDurationNs']) for r in rows if k in r['Name'])/1e4 for k in ('k_colour_bwd<true>','k_sdfnet4_bwd<8, 4, 3, true>','k_sdfnet4_bwd<4, 8, 1, true>','k_emit_gemm','k_emit_reduce','k_adam_table')}
    print(f.split('/')[-1], 'total %.3f ms/it' % (tot/1e7), {k: round(v,1) for k,v in pick.items()})
PY
