#!/bin/bash
# One parametrised GPU-box script (replaces the per-call run logs of earlier rounds):  gpurun -- 'bash tools/gpu/run.sh <tag> <step> [<step> ..]'
# Everything a step writes goes to gpurun_out/<tag>/ ; summaries worth keeping are copied into profiles/ by hand afterwards.
cd ${GRAFT_REPO_ROOT:-.}
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
for step in "$@"; do
  echo "=== $step"
  case $step in
    tests)        timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log ;;
    testsall)     timeout 2400 python -m pytest tests -m gpu -q > $O/tests_all.log 2>&1; echo "rc=$?" >> $O/tests_all.log; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_all.log | tail -30 ;;
    ab:*)         # ab:<ENV>: bench.py's tracker line with ENV=0 / ENV=1 alternating, three rounds (per-kernel event times in the lines)
                  V=${step#ab:}; for r in 1 2 3; do for f in 0 1; do
                    env $V=$f timeout 600 python bench.py --no-cpu-baseline --no-mapping --no-dropin --no-precision-modes --no-small-shapes \
                        > $O/ab_${V}_${f}_$r.json 2>/dev/null
                    python - $O/ab_${V}_${f}_$r.json $V=$f <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["roofline"]["all_kernels_us"]
print(sys.argv[2], "ms", d["ms_per_step"], {n: v for n, v in k.items() if "sdfnet" in n or "sampler_sdf" in n})
PY
                  done; done ;;
    pmcab:*)      # pmcab:<ENV>: SQ counters of the quad kernels with ENV=0 / ENV=1 (one counter pass each; eager launches)
                  V=${step#pmcab:}; R=$(pwd)
                  BE="python $R/bench.py --no-cpu-baseline --no-mapping --no-dropin --no-precision-modes --no-small-shapes --steps 20 --warmup 5 --prewarm-s 0 --prewarm-steps 0 --no-graph"
                  for f in 0 1; do
                    (cd /tmp && env $V=$f timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU \
                        --output-format csv -d /tmp/pmcab_$f -- $BE > /tmp/pmcab_$f.log 2>&1)
                    python tools/pmc_summary.py /tmp/pmcab_$f | grep -E "kernel|sdfnet4" > $O/pmc_${V}_$f.csv
                    echo "$V=$f"; cat $O/pmc_${V}_$f.csv | cut -c1-150
                  done ;;
    chunks)       # ray chunks on forked streams inside the graph (KernelTracker(chunks=)): 1 vs 2 at the shipped sample count and the headline's
                  for smp in 98 128; do for c in 1 2 1 2; do
                    timeout 300 python bench.py --samples $smp --chunks $c --no-cpu-baseline --no-mapping --no-precision-modes --no-dropin --no-small-shapes \
                        2>/dev/null > $O/chunks_${smp}_$c.json
                    python -c "import json,sys; d=json.loads(open(sys.argv[1]).readline()); print('samples', sys.argv[2], 'chunks', sys.argv[3], d['ms_per_step'])" $O/chunks_${smp}_$c.json $smp $c
                  done; done ;;
    abtags:*)     # abtags:<tag,tag,..>: tools/ab_kernels.py (fp32) for the product library and tagged side-by-side builds, two rounds
                  for rnd in 1 2; do for tag in "" $(echo ${step#abtags:} | tr ',' ' '); do
                    NSA_LIB_TAG=$tag timeout 300 python tools/ab_kernels.py --steps 60 2>/dev/null > $O/abt_${tag:-product}_$rnd.json
                    python -c "import json,sys; d=json.loads(open(sys.argv[1]).readline()); print(sys.argv[2] or 'product', 'graph ms', d.get('graph_ms'), {n.replace('k_',''): v for n, v in d['kernels_us'].items()})" $O/abt_${tag:-product}_$rnd.json "$tag"
                  done; done ;;
    warmab)       # the driver's command (--steps 20 --warmup 5) under different untimed pre-warm recipes: does the short run reach the long run's clocks?
                  for rnd in 1 2; do for v in "1.0 100" "0 100" "1.0 400" "1.0 1000" "0 1000" "0.3 2000" "3.0 100"; do set -- $v
                    timeout 300 python bench.py --steps 20 --warmup 5 --prewarm-s $1 --prewarm-steps $2 --no-cpu-baseline --no-mapping --no-dropin --no-precision-modes --no-small-shapes \
                        2>/dev/null > $O/warm_${1}_${2}_$rnd.json
                    python -c "import json,sys; d=json.loads(open(sys.argv[1]).readline()); print('prewarm_s', sys.argv[2], 'prewarm_steps', sys.argv[3], 'ms', d['ms_per_step'], 'sampler', d['roofline']['all_kernels_us']['k_sampler_sdf'])" $O/warm_${1}_${2}_$rnd.json $1 $2
                  done; done
                  timeout 300 python bench.py --no-cpu-baseline --no-mapping --no-dropin --no-precision-modes --no-small-shapes 2>/dev/null > $O/warm_default.json
                  python -c "import json,sys; d=json.loads(open(sys.argv[1]).readline()); print('default 200/20 ms', d['ms_per_step'])" $O/warm_default.json ;;
    ramp)         # tools/diag_step_ramp.py: per-iteration device times and host call times inside a short timed region
                  for v in "20 --rounds 4" "20 --fresh --rounds 4" "20 --fresh --reset --rounds 4" "20 --fresh --reset --gemm-s 1 --gc-off --rounds 4" "200 --fresh --rounds 3"; do set -- $v
                    F=$O/ramp_$(echo $v | tr -d ' -').json
                    timeout 300 python tools/diag_step_ramp.py --steps $v 2>/dev/null > $F
                    echo "-- $v"
                    python - $F <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
print("steps", d["steps"], "idle_ms", d["idle_ms"])
for r in d["rounds"]:
    g, h = r["gpu_step_us"], r["host_step_us"]
    print("  wall/step", r["wall_ms_per_step"], "queued_after_us", r["queued_after_us"], "region_us", r["region_us"], "gpu_ev_span", r["gpu_first_to_last_event_us"],
          "| gpu first 6", g[:6], "last 3", g[-3:], "| host first 4", h[:4], "median", sorted(h)[len(h) // 2])
PY
                  done ;;
    slamtest:*)   # slamtest:<n>: the mini-SLAM tests n times (their ATE scatters from run to run: atomics in the mapping step); prints the ATE lines
                  for i in $(seq 1 ${step#slamtest:}); do
                    timeout 600 python -m pytest tests/test_sequence_gpu.py -m gpu -q -k mini_slam -s 2>&1 | grep -E "mini-SLAM \[|passed|failed" | cut -c1-150
                  done ;;
    ablate)       # timing-only ablation builds (wrong numbers): product vs Softplus-free vs Softplus- and PE-free, fp32 and bf16 operands
                  for prec in fp32 bf16; do for tag in "" spfree vfree "" spfree vfree; do
                    NSA_LIB_TAG=$tag timeout 300 python tools/ab_kernels.py --precision $prec --steps 60 2>/dev/null > $O/abl_${prec}_${tag:-product}.json
                    python -c "import json,sys; d=json.loads(open(sys.argv[1]).readline()); k=d.get('kernels_us', d); print(sys.argv[2], sys.argv[3] or 'product', 'graph ms', d.get('graph_ms'), {n: v for n, v in k.items() if isinstance(v, (int, float)) and ('sampler_sdf' in n or 'sdfnet' in n)})" $O/abl_${prec}_${tag:-product}.json $prec "$tag"
                  done; done ;;
    tests:*)      timeout 1500 python -m pytest ${step#tests:} -m gpu -x -q > $O/tests_sel.log 2>&1; echo "rc=$?" >> $O/tests_sel.log; tail -15 $O/tests_sel.log ;;
    smoke)        timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 ;;
    bench)        timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?"; cut -c1-600 $O/bench.json ;;
    bench:*)      timeout 900 python bench.py ${step#bench:} > $O/bench_opt.json 2> $O/bench_opt.err; echo "rc=$?"; cut -c1-600 $O/bench_opt.json ;;
    seq7)         timeout 1200 python tools/synthetic_sequence.py --conf 7scenes > $O/sequence_ate_7scenes.json 2> $O/seq7.err; echo "rc=$?"; tail -3 $O/seq7.err ;;
    seqA)         timeout 1500 python tools/synthetic_sequence.py --conf azure --n-samples 158 --oracle-frames 0 > $O/sequence_ate_azure.json 2> $O/seqA.err; echo "rc=$?"; tail -3 $O/seqA.err ;;
    seqR)         timeout 1200 python tools/synthetic_sequence.py --conf replica > $O/sequence_ate_replica.json 2> $O/seqR.err; echo "rc=$?"; tail -3 $O/seqR.err ;;
    slam7)        timeout 2400 python tools/synthetic_sequence.py --conf 7scenes --slam --schedule reference --none-grad skip,zeros \
                      --engines fused:11+12+13+14+15,composed:12 > $O/slam_ate_7scenes.json 2> $O/slam7.err; echo "rc=$?"; tail -3 $O/slam7.err ;;
    slamR)        timeout 2400 python tools/synthetic_sequence.py --conf replica --slam --schedule reference --none-grad skip,zeros \
                      --engines fused:11+12+13+14+15 > $O/slam_ate_replica.json 2> $O/slamR.err; echo "rc=$?"; tail -3 $O/slamR.err ;;
    slamA)        # mini-SLAM on frames of the closed-form textured box room (no network rendered them): both families' conf + trajectory
                  timeout 2400 python tools/synthetic_sequence.py --conf 7scenes --slam --analytic --schedule reference --none-grad skip,zeros \
                      --engines fused:11+12+13+14+15,composed:12 > $O/slam_ate_analytic_7scenes.json 2> $O/slamA7.err; echo "rc=$?"; tail -2 $O/slamA7.err
                  timeout 2400 python tools/synthetic_sequence.py --conf replica --slam --analytic --schedule fine \
                      --engines fused:11+12+13+14+15,composed:12 > $O/slam_ate_analytic_replica.json 2> $O/slamAR.err; echo "rc=$?"; tail -2 $O/slamAR.err ;;
    slam7fine)    timeout 2400 python tools/synthetic_sequence.py --conf 7scenes --slam --schedule fine --engines fused:11+12+13,composed:12 \
                      > $O/slam_ate_7scenes_fine.json 2> $O/slam7fine.err; echo "rc=$?"; tail -3 $O/slam7fine.err ;;
    profile)      bash tools/profile_round.sh > $O/profile.log 2>&1; tail -5 $O/profile.log | cut -c1-200 ;;
    profile_map)  bash tools/profile_mapping.sh > $O/profile_map.log 2>&1; tail -4 $O/profile_map.log | cut -c1-200 ;;
    sh:*)         bash -c "${step#sh:}" ;;
    *)            echo "unknown step $step" ;;
  esac
done
