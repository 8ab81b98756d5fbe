#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel stats + atomic-request counters of the mapping iteration (bench.py
# --only-mapping: the shipped Replica objective) -> gpurun_out/prof/mapping_*.   usage: tools/profile_mapping.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --only-mapping 7 --no-cpu-baseline"
BE="python $R/bench.py --only-mapping 2 --no-cpu-baseline"
rm -rf /tmp/mk /tmp/m1 /tmp/m2
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mk -- $B > /tmp/mk.log 2>&1
cp $(find /tmp/mk -name "*kernel_stats.csv" | head -1) $OUT/mapping_kernel_stats.csv
# counter names as in profiles/r01_mapping_pmc_per_kernel.csv (an unknown name makes rocprofv3 abort after the full timeout)
if [ "${1:-}" != "--stats-only" ]; then
timeout 240 rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum --output-format csv -d /tmp/m1 -- $BE > /tmp/m1.log 2>&1
python $R/tools/pmc_summary.py /tmp/m1 > $OUT/mapping_pmc_per_kernel.csv
(cd $R && python -c "import json, bench; print(json.dumps(bench.mapping_signature()))") > $OUT/mapping_pmc_meta.json
fi
head -25 $OUT/mapping_kernel_stats.csv | cut -c1-150
wc -l $OUT/mapping_pmc_per_kernel.csv; tail -3 /tmp/m1.log
