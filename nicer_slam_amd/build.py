"""Ahead-of-time build of libnicer_slam_amd.so (hipcc, gfx950 only; cross-compiles without a GPU).

Replaces the reference's import-time JIT (code/hashencoder/backend.py:30-42).  Objects go to
``nicer_slam_amd/build/``, the library to ``nicer_slam_amd/lib/`` -- both git-ignored, both travel to the
GPU box with the tree.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# NSA_BUILD_TAG=<tag> (+ NSA_EXTRA_HIPCC_FLAGS): a side-by-side experiment build (tools/ab_kernels.py), never the product
TAG = os.environ.get("NSA_BUILD_TAG", "")
OBJ = os.path.join(HERE, "build" + ("_" + TAG if TAG else ""))
LIB = os.path.join(HERE, "lib", "libnicer_slam_amd" + ("_" + TAG if TAG else "") + ".so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize is a CORRECTNESS flag here, not a tuning knob.  With SLP vectorisation (on at -O2/-O3) hipcc (ROCm 7.2)
# packs adjacent fp32 multiplies / adds / fmas of the per-point code into v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, and the
# MFMA kernels then return run-to-run DIFFERENT values on MI355X at ~1e-4 of the points (always lanes >= 16 of a wave, errors
# up to 3e-3 in sdf) -- the same binary, the same inputs (tools/diag_determinism.py; -O1 and -fno-slp-vectorize builds are
# bit-reproducible and equal the CPU oracle, every -O2/-O3 build with packed fp32 math is not; tests/test_tiling_gpu.py holds the
# regression).  The packed forms are also slower beside MFMAs (MI355X_MICROARCH.md, per-instruction constants).
FLAGS = ["--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function"] + os.environ.get("NSA_EXTRA_HIPCC_FLAGS", "").split()
# NSA_EXP_SLP=1 with a NSA_BUILD_TAG: a side-by-side build WITH the SLP vectoriser, for the hazard experiments of
# tools/slp_hazard_experiments.sh only (the untagged product library can never be built this way, see _check_flags)
if TAG and os.environ.get("NSA_EXP_SLP") == "1":
    FLAGS.remove("-fno-slp-vectorize")


def _check_flags():
    """The product library (no NSA_BUILD_TAG) must be the product: experiment macros (NSA_X_*: the profiling instrumentation of
    tools/ts_profile*.py, tools/slot_timeline.py; NSA_EXP_* / NSA_ABL_*: none left in the sources since round 4) change kernels
    and are only accepted for a tagged side-by-side build; and the correctness
    flag -fno-slp-vectorize (above) cannot be dropped or overridden in any build."""
    extra = os.environ.get("NSA_EXTRA_HIPCC_FLAGS", "").split()
    bad = [f for f in extra if any(f.startswith("-D" + p) for p in ("NSA_ABL_", "NSA_EXP_", "NSA_X_"))]
    if bad and not TAG:
        raise SystemExit(f"build.py: {bad} are experiment macros; set NSA_BUILD_TAG=<tag> for a side-by-side build "
                         "(the untagged library is the product)")
    if any(f in ("-fslp-vectorize", "-fvectorize") for f in extra) or ("-fno-slp-vectorize" not in FLAGS and not TAG):
        raise SystemExit("build.py: -fno-slp-vectorize is a correctness flag on gfx950 (see the comment above FLAGS)")


OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")


def isa_check(lib):
    """Build-time ISA check of the PRODUCT library: no packed-fp32 arithmetic (v_pk_mul/add/fma_f32) in any gfx950 code object.
    Those instructions are what the SLP vectoriser makes of adjacent fp32 math; with them the quad-tiling MFMA kernels return
    run-to-run different values on MI355X (tools/slp_hazard_experiments.sh).  -fno-slp-vectorize keeps them out; this
    check makes a toolchain or flag change that brings them back a build failure instead of a silent numerical one.
    Returns {instruction: count} of the offenders (empty = clean); skipped (None) when llvm-objdump is not available."""
    import re
    import shutil
    import tempfile
    if not os.path.exists(OBJDUMP):
        return None
    tmp = tempfile.mkdtemp(prefix="nsa_isa_")
    try:
        copy = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, copy)
        subprocess.run([OBJDUMP, "--offloading", copy], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        found = {}
        n_objs = 0
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            n_objs += 1
            dis = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            for m in re.finditer(r"\b(v_pk_(?:mul|add|fma)_f32)\b", dis):
                found[m.group(1)] = found.get(m.group(1), 0) + 1
        if n_objs == 0:
            raise SystemExit(f"build.py: isa_check found no gfx950 code object in {lib}")
        return found
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# Experiment kernels that no default path selects: the two wave-specialised samplers of round 4 (tile codes 96 / 97,
# bit-identical to k_sampler_sdf and at parity or slower).  They are NOT part of the product library; a tagged side-by-side build
# made with NSA_X_WS=1 compiles them in (tools/pmc_ws_sampler.sh, tools/ts_profile_ws.py, tests/test_tiling_gpu.py when that
# library is loaded through NSA_LIB_TAG).
EXPERIMENT_SOURCES = ("render_sampler_ws.hip", "render_sampler_sys.hip")
WITH_WS = bool(TAG) and os.environ.get("NSA_X_WS") == "1"
if WITH_WS:
    FLAGS.append("-DNSA_X_WS_SAMPLERS")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=False):
    _check_flags()
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    all_srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    srcs = [s for s in all_srcs if WITH_WS or os.path.basename(s) not in EXPERIMENT_SOURCES]
    hdrs = (sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + sorted(glob.glob(os.path.join(CSRC, "*.inc")))
            + [os.path.join(HERE, "..", "include", "nicer_slam_amd.h")])
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        # (every .hip counts as a dependency of every object: the *_bf16.hip units #include other .hip files)
        if force or _stale(o, all_srcs + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        if not TAG:                                   # (experiment builds may contain anything)
            bad = isa_check(LIB)
            if bad:
                os.remove(LIB)
                raise SystemExit(f"build.py: packed-fp32 arithmetic in the product library {bad}: these make the MFMA kernels "
                                 "irreproducible on gfx950 (DESIGN.md 4.1); check that -fno-slp-vectorize reached every compile")
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
