#!/bin/bash
# mapping iteration A/B: table-gradient clearing policy x Morton key bits (bench.py --only-mapping, no profiler)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5x; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_mapping_gpu.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
B="python bench.py --only-mapping 10 --no-cpu-baseline"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/map_$name.json 2> $O/map_$name.err; echo "$name rc=$? $(python -c "import json;print(json.load(open('$O/map_$name.json')).get('rays_per_s'))" 2>/dev/null)"; }
run acquire30 NSA_X=0
# run async30 NSA_TABLE_GRAD_CLEAR=async      (policy removed after this run)
run fused30 NSA_TABLE_GRAD_CLEAR=fused
run autograd30 NSA_TABLE_GRADS=autograd
run acquire27 NSA_MORTON_BITS=27
run acquire24 NSA_MORTON_BITS=24
run acquire21 NSA_MORTON_BITS=21
run acquire18 NSA_MORTON_BITS=18
run acquire30b NSA_X=0
