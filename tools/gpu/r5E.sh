#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5E; mkdir -p $O
(for m in linear1 nt ntl nts nt linear1; do echo "== adam $m"; NSA_ADAM_GRID=$m timeout 200 python tools/ab_adam.py 2>/dev/null | tail -2; done
for g in 1 9 1 9; do NSA_FILL_GROUPS=$g timeout 120 python tools/micro/fill_bench.py 2>/dev/null | tail -1; done) | tee $O/adam_fill_nt.txt
