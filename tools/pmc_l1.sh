#!/bin/bash
# Run ON THE GPU BOX: vector-memory path counters (TA = address / coalescing unit, TCP = vector L1, TD = data return) of the tracking
# iteration's kernels -> gpurun_out/prof/pmc_l1_per_kernel.csv.  Is a kernel bound by the per-CU L1 path rather than by issue or HBM?
# (counter names from `rocprofv3 -L` on gfx950; an unknown name makes rocprofv3 hang until the timeout)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BE="python $R/bench.py --no-cpu-baseline --no-mapping --no-dropin --no-precision-modes --no-small-shapes --steps 6 --warmup 2 --prewarm-s 0 --prewarm-steps 0 --no-graph"
rm -rf /tmp/l1a /tmp/l1b /tmp/l1c
timeout 70 rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE TA_TOTAL_WAVEFRONTS_sum --output-format csv -d /tmp/l1a -- $BE > /tmp/l1a.log 2>&1
timeout 70 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d /tmp/l1b -- $BE > /tmp/l1b.log 2>&1
timeout 70 rocprofv3 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum --output-format csv -d /tmp/l1c -- $BE > /tmp/l1c.log 2>&1
python $R/tools/pmc_summary.py /tmp/l1a /tmp/l1b /tmp/l1c > $OUT/pmc_l1_per_kernel.csv
wc -l $OUT/pmc_l1_per_kernel.csv; grep "sampler_sdf\|colour_fwd" $OUT/pmc_l1_per_kernel.csv
tail -2 /tmp/l1a.log /tmp/l1b.log /tmp/l1c.log | cut -c1-160
