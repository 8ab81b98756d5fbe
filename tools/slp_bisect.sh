#!/bin/bash
# bisect the SLP (v_pk_*_f32) irreproducibility of the quad-tiling kernels on OUR staging / LDS code.
# The affected kernels (tools/diag_slp.py: k_sdf4_points / k_sampler4_sdf) keep their weights RESIDENT in LDS for the whole launch
# (one copy, one barrier at kernel start, no refill), so there is no stage buffer to race on; what an LDS-side cause could still be:
#   slpnop          control: SLP on + s_nop 2 in front of every instruction   (round 4: ~950-2200 of 640 000 points differ between runs)
#   slpnop_glb      + weight fragments streamed from GLOBAL memory (gemm16_glb): no ds_read_b128 between the MFMAs
#   slpnop_fence    + s_waitcnt lgkmcnt(0) after the fragment reads of every chunk: every LDS read has returned before its MFMAs issue
#   slpnop_geom     + level geometry read field by field (volatile): no ds_read2_b32 merges in the grid blend's neighbourhood
#   tools/slp_bisect.sh          build the four libraries here (hipcc cross-compiles)
#   tools/slp_bisect.sh run      on the GPU box: tools/diag_slp.py on each
cd "$(dirname "$0")/.."
TAGS="slpnop slpnop_glb slpnop_fence slpnop_geom"
if [ "${1:-}" == "run" ]; then
  for t in "" $TAGS; do echo "=== ${t:-product}"; NSA_LIB_TAG=$t timeout 200 python tools/diag_slp.py 2>&1 | grep -v "Warn\|warn" | grep "lib\|differ\|spread\|RIGHT"; done
  exit 0
fi
P="-mllvm -amdgpu-snop-padding=2"
NSA_EXP_SLP=1 NSA_BUILD_TAG=slpnop NSA_EXTRA_HIPCC_FLAGS="$P" python -m nicer_slam_amd.build > /dev/null 2>&1 &
NSA_EXP_SLP=1 NSA_BUILD_TAG=slpnop_glb NSA_EXTRA_HIPCC_FLAGS="$P -DNSA_X_GLB_WEIGHTS" python -m nicer_slam_amd.build > /dev/null 2>&1 &
wait
NSA_EXP_SLP=1 NSA_BUILD_TAG=slpnop_fence NSA_EXTRA_HIPCC_FLAGS="$P -DNSA_X_LDS_FENCE" python -m nicer_slam_amd.build > /dev/null 2>&1 &
NSA_EXP_SLP=1 NSA_BUILD_TAG=slpnop_geom NSA_EXTRA_HIPCC_FLAGS="$P -DNSA_X_GEOM_VOLATILE" python -m nicer_slam_amd.build > /dev/null 2>&1 &
wait
ls -la nicer_slam_amd/lib/
