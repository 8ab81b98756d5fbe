#!/bin/bash
# Run ON THE GPU BOX: matrix-pipe co-execution counters of the three sampler forms (shipped two-tile program, matrix waves on loan = 96,
# systolic layer engines = 97) -> gpurun_out/prof/pmc_ws_sampler.csv   (SQ_VALU_MFMA_COEXEC_CYCLES / ..._BUSY_CYCLES)
# needs the experiment build:  NSA_BUILD_TAG=ws NSA_X_WS=1 python -m nicer_slam_amd.build   and   export NSA_LIB_TAG=ws  (the product
# library does not contain tile codes 96 / 97 since round 5)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BE="python $R/bench.py --no-cpu-baseline --no-mapping --no-dropin --no-precision-modes --no-small-shapes --steps 6 --warmup 2 --prewarm-s 0 --prewarm-steps 0 --no-graph"
: > $OUT/pmc_ws_sampler.csv
for t in 64 96 97; do
  rm -rf /tmp/ws$t
  NSA_SAMPLER_TILE=$t timeout 80 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
      --output-format csv -d /tmp/ws$t -- $BE > /tmp/ws$t.log 2>&1
  echo "# NSA_SAMPLER_TILE=$t" >> $OUT/pmc_ws_sampler.csv
  python $R/tools/pmc_summary.py /tmp/ws$t | grep -i "kernel,counter\|sampler" >> $OUT/pmc_ws_sampler.csv
done
cat $OUT/pmc_ws_sampler.csv
