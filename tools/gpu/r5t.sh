#!/bin/bash
# round 5, GPU call T: PMC pass over the bf16 bench (resident quad kernels), with logs.
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
O=$R/gpurun_out/r5t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BE="python $R/bench.py --precision bf16 --no-cpu-baseline --no-mapping --no-dropin --no-precision-modes --steps 20 --warmup 5 --prewarm-s 0 --prewarm-steps 0 --no-graph"
(timeout 200 $BE) > $O/plain.json 2> $O/plain_err.log; echo "plain rc=$?" >> $O/plain_err.log
rm -rf /tmp/p1 /tmp/p2
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d /tmp/p1 -- $BE > $O/p1.log 2>&1; echo "p1 rc=$?" >> $O/p1.log
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT TCC_MISS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d /tmp/p2 -- $BE > $O/p2.log 2>&1; echo "p2 rc=$?" >> $O/p2.log
python $R/tools/pmc_summary.py /tmp/p1 /tmp/p2 > $O/bf16_pmc_per_kernel.csv
cd $R
tail -3 $O/plain_err.log; tail -5 $O/p1.log | cut -c1-300; wc -l $O/bf16_pmc_per_kernel.csv; grep "sampler_sdf\|pair_res\|bwd_res" $O/bf16_pmc_per_kernel.csv | cut -c1-160 | head -45
