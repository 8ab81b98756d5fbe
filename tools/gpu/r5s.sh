#!/bin/bash
# round 5, GPU call S: rocprofv3 kernel stats of the mapping iteration (final code) and a PMC pass over the bf16 bench.
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
O=$R/gpurun_out/r5s; mkdir -p $O
timeout 600 bash tools/profile_mapping.sh --stats-only > $O/profile_mapping.log 2>&1
cp gpurun_out/prof/mapping_kernel_stats.csv $O/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
BE="python $R/bench.py --precision bf16 --no-cpu-baseline --no-mapping --no-dropin --no-precision-modes --steps 20 --warmup 5 --prewarm-s 0 --prewarm-steps 0 --no-graph"
rm -rf /tmp/p1 /tmp/p2
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d /tmp/p1 -- $BE > /tmp/p1.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT TCC_MISS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d /tmp/p2 -- $BE > /tmp/p2.log 2>&1
python $R/tools/pmc_summary.py /tmp/p1 /tmp/p2 > $O/bf16_pmc_per_kernel.csv
cd $R
head -30 $O/mapping_kernel_stats.csv | cut -c1-140; wc -l $O/bf16_pmc_per_kernel.csv; grep "sampler_sdf\|fwd_pair_res\|bwd_res" $O/bf16_pmc_per_kernel.csv | cut -c1-200 | head -40
