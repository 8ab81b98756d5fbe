#!/bin/bash
# Which class of hazard makes SLP-vectorised (v_pk_*_f32) builds of the MFMA kernels irreproducible on gfx950?  
# Builds side-by-side libraries WITH the SLP vectoriser plus one counter-measure each (run here, hipcc cross-compiles), then
#   tools/slp_hazard_experiments.sh run      on the GPU box: tools/diag_determinism.py on every variant.
#     slp        plain -O3 (SLP on)                                   -> expected irreproducible
#     slpwc0     + every s_waitcnt forced to 0                        -> reproducible <=> a counter (vmcnt / lgkmcnt) the compiler under-waits
#     slpnop     + s_nop 2 in front of EVERY instruction              -> reproducible <=> missing wait states between two instructions (a VALU / MFMA / trans hazard)
#     slppad     + MFMA padding ratio 100                             -> reproducible <=> specific to MFMA neighbourhoods
cd "$(dirname "$0")/.."
if [ "${1:-}" == "run" ]; then
  for t in "" slp slpwc0 slpnop slppad; do NSA_LIB_TAG=$t timeout 120 python tools/diag_determinism.py 2>&1 | grep -v Warn | tail -1; done
  exit 0
fi
NSA_EXP_SLP=1 NSA_BUILD_TAG=slp python -m nicer_slam_amd.build > /dev/null 2>&1 &
NSA_EXP_SLP=1 NSA_BUILD_TAG=slpwc0 NSA_EXTRA_HIPCC_FLAGS="-mllvm -amdgpu-waitcnt-forcezero" python -m nicer_slam_amd.build > /dev/null 2>&1 &
wait
NSA_EXP_SLP=1 NSA_BUILD_TAG=slpnop NSA_EXTRA_HIPCC_FLAGS="-mllvm -amdgpu-snop-padding=2" python -m nicer_slam_amd.build > /dev/null 2>&1 &
NSA_EXP_SLP=1 NSA_BUILD_TAG=slppad NSA_EXTRA_HIPCC_FLAGS="-mllvm -amdgpu-mfma-padding-ratio=100" python -m nicer_slam_amd.build > /dev/null 2>&1 &
wait
ls -la nicer_slam_amd/lib/
