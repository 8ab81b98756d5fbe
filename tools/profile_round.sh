#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel stats + PMC passes of bench.py -> gpurun_out/prof/
# usage: tools/profile_round.sh   (from the repo root)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-mapping --no-dropin --no-precision-modes --no-small-shapes --steps 200 --warmup 20"
# counter passes serialise the kernels and are ~50x slower: 20 eager steps are plenty for per-launch means
BE="python $R/bench.py --no-cpu-baseline --no-mapping --no-dropin --no-precision-modes --no-small-shapes --steps 20 --warmup 5 --prewarm-s 0 --prewarm-steps 0 --no-graph"
(timeout 400 $B) > $OUT/bench_line.json 2> /tmp/b.err || true
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- $B > /tmp/ks.log 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/f1 -- $BE > /tmp/f1.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/f2 -- $BE > /tmp/f2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d /tmp/f3 -- $BE > /tmp/f3.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT TCC_MISS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d /tmp/f4 -- $BE > /tmp/f4.log 2>&1
python $R/tools/pmc_summary.py /tmp/f1 /tmp/f2 /tmp/f3 /tmp/f4 > $OUT/pmc_per_kernel.csv
python $R/tools/traffic_json.py $OUT/pmc_per_kernel.csv > $OUT/hbm_traffic.json
tail -1 $OUT/bench_line.json | cut -c1-200
wc -l $OUT/pmc_per_kernel.csv
cat $OUT/hbm_traffic.json
