#!/bin/bash
# fill micro-benchmark, mapping tests, interleaved A/B, mapping profile (kernel stats + atomic-request counters)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5A; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for c in 1024 2048 4096 8192; do NSA_FILL_BLOCKS=$c timeout 120 python tools/micro/fill_bench.py 2>/dev/null | tail -1; done | tee $O/fill_bench.txt
timeout 900 python -m pytest tests/test_mapping_gpu.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -3 $O/tests.log
timeout 600 python tools/ab_mapping.py 30 3 > $O/ab.txt 2> $O/ab.err; head -6 $O/ab.txt | cut -c1-150
bash tools/profile_mapping.sh > $O/profile.log 2>&1
cp gpurun_out/prof/mapping_kernel_stats.csv gpurun_out/prof/mapping_pmc_per_kernel.csv gpurun_out/prof/mapping_pmc_meta.json $O/
tail -4 $O/profile.log | cut -c1-200
