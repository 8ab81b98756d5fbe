#!/bin/bash
# mapping iteration: in-tree weight norm / table-gradient buffers / radix sort / emit rows -- tests, A/B timings, kernel stats
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5w; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_mapping_gpu.py tests/test_pack_gpu.py tests/test_model_gpu.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -15 $O/tests.log
B="python bench.py --only-mapping 10 --no-cpu-baseline"
for v in default bits24 autograd; do
  case $v in
    default) E="";;
    bits24) E="NSA_MORTON_BITS=24";;
    autograd) E="NSA_TABLE_GRADS=autograd";;
  esac
  env $E timeout 300 $B > $O/map_$v.json 2> $O/map_$v.err; echo "$v rc=$?"; cut -c1-400 $O/map_$v.json
done
bash tools/profile_mapping.sh --stats-only > $O/profile.log 2>&1
cp gpurun_out/prof/mapping_kernel_stats.csv $O/
head -30 $O/mapping_kernel_stats.csv | cut -c1-160
