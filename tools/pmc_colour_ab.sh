#!/bin/bash
# (counter names must be ones rocprofv3 knows on gfx950 -- TCC_HIT, TCC_MISS, FETCH_SIZE, WRITE_SIZE ...: an unknown name makes it hang until the timeout)
# Run ON THE GPU BOX: L2 request counters of k_colour_fwd with and without the x-pair gather (NSA_COLOUR_XPAIR) -> gpurun_out/prof/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/cx$v
  NSA_COLOUR_XPAIR=$v timeout 90 rocprofv3 --pmc TCC_HIT TCC_MISS --output-format csv -d /tmp/cx$v -- python $R/tools/ab_kernels.py --steps 3 --no-graph-leg > /tmp/cx$v.log 2>&1
  python $R/tools/pmc_summary.py /tmp/cx$v | grep "colour_fwd\|kernel," > $OUT/colour_xpair_$v.csv
done
tail -n +1 $OUT/colour_xpair_0.csv $OUT/colour_xpair_1.csv
