#!/bin/bash
# Run ON THE GPU BOX (via gpurun): PMC counters of the tracking iteration's kernels (eager replay), one summary CSV.
# usage: tools/profile_ab.sh <out-name>   (NSA_SDF_TILE / NSA_LIB_TAG select the build / tiling)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C="python $R/tools/ab_kernels.py --no-graph-leg --steps 10"
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d /tmp/p1 -- $C > /tmp/p1.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES --output-format csv -d /tmp/p2 -- $C > /tmp/p2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT TCC_MISS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d /tmp/p3 -- $C > /tmp/p3.log 2>&1
python $R/tools/pmc_summary.py /tmp/p1 /tmp/p2 /tmp/p3 > $OUT/$1.csv
wc -l $OUT/$1.csv; tail -3 /tmp/p1.log
