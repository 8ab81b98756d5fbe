"""The C-ABI library loads without a GPU and exports every symbol include/nicer_slam_amd.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "nicer_slam_amd.h")).read()
    declared = set(re.findall(r"\b(nsa_[a-z0-9_]+)\s*\(", hdr))
    assert {"nsa_hash_encode_forward", "nsa_hash_encode_backward", "nsa_hash_encode_second_backward"} <= declared
    lib = ctypes.CDLL(os.path.join(ROOT, "nicer_slam_amd", "lib", "libnicer_slam_amd.so"))
    for name in sorted(declared):
        assert hasattr(lib, name), name
    lib.nsa_strerror.restype = ctypes.c_char_p
    assert lib.nsa_strerror(1) == b"GridEncoding: C must be 1, 2, 4, or 8."
    assert lib.nsa_version() >= 1


def test_native_binding_lists_all_exports():
    from nicer_slam_amd import _native
    hdr = open(os.path.join(ROOT, "include", "nicer_slam_amd.h")).read()
    declared = set(re.findall(r"\b(nsa_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_native.EXPORTS)


def test_argument_validation_needs_no_gpu():
    """Every entry point checks its arguments before touching the device: NULL pointers and inconsistent sizes come back
    as NSA_EBADARG with a message, unsupported shapes as NSA_EUNSUPPORTED_*; none of these calls launches anything."""
    import ctypes
    from nicer_slam_amd._native import lib, check, GridDesc, PointsDesc
    NSA_EBADARG, NSA_EUNSUPPORTED_C, NSA_EUNSUPPORTED_NET = 4, 1, 5
    assert lib.nsa_version() >= 1
    assert lib.nsa_strerror(0) == b"ok"
    assert lib.nsa_strerror(NSA_EUNSUPPORTED_C) == b"GridEncoding: C must be 1, 2, 4, or 8."      # hashencoder.cu:637
    assert b"bad argument" in lib.nsa_strerror(NSA_EBADARG)
    with pytest.raises(RuntimeError, match="bad argument"):
        check(NSA_EBADARG)
    pts = PointsDesc(None, None, None, None, 64, 0, None)                # neither rays nor points
    grid = GridDesc(None, None, 4, 8, 0.0, 32, 1.0, 1)
    assert lib.nsa_sdfnet_forward(ctypes.byref(pts), ctypes.byref(grid), None, 0, None, None, None, None) == NSA_EBADARG
    assert lib.nsa_sdfnet_backward(None, None, None, None, None, None, 0, None, None) == NSA_EBADARG
    assert lib.nsa_sdfnet_backward_params(None, None, None, None, None, None, 0, None, None, None, 0, None) == NSA_EBADARG
    assert lib.nsa_colour_forward(None, None, None, None, None, None, None, None) == NSA_EBADARG
    assert lib.nsa_update_voxels(None, None, 64, None) == NSA_EBADARG
    assert lib.nsa_morton_keys(None, None, None) == NSA_EBADARG
    assert lib.nsa_adam_table_step(None, None, None, None, 16, 1, 0.1, 0.9, 0.99, 1e-15, None) == NSA_EBADARG
    assert lib.nsa_draw_picks(None, 640, 32, 8, 98, None, None, None) == NSA_EBADARG
    assert lib.nsa_track_head(None, None, None, 0, None, None, None, None, None) == NSA_EBADARG
    assert lib.nsa_sdf_points(None, 8, None, None, None, None, None, None) == NSA_EBADARG
    # round-2 entry points: weight-gradient GEMM over emission rows, fused loss terms
    from nicer_slam_amd._native import LossDesc
    a_rows = (ctypes.c_uint32 * 2)(0, 0)
    assert lib.nsa_emit_gemm(None, 4096, 1, a_rows, a_rows, 64, 64, 1, None, None, None) == NSA_EBADARG          # NULL buffers
    fake = ctypes.c_void_p(4096)                                                      # never dereferenced: rejected on sizes
    assert lib.nsa_emit_gemm(fake, 4096, 3, a_rows, a_rows, 64, 64, 1, fake, fake, None) == NSA_EBADARG          # pairs > 2
    assert lib.nsa_emit_gemm(fake, 4096, 1, a_rows, a_rows, 65, 64, 1, fake, fake, None) == NSA_EBADARG          # M > 64
    assert lib.nsa_emit_gemm(fake, 1000, 1, a_rows, a_rows, 64, 64, 1, fake, fake, None) == NSA_EBADARG          # ld not a multiple of 256
    assert lib.nsa_emit_gemm_workspace(8192, 64, 130, 1) == (8192 // 256) * 64 * 131
    assert lib.nsa_slam_loss(None, None, None) == NSA_EBADARG
    assert lib.nsa_slam_loss(ctypes.byref(LossDesc()), fake, None) == NSA_EBADARG     # zero sizes / NULL tensors
    assert lib.nsa_slam_loss_workspace(8, 1024, 0) > 8 * 1024
    # round-3 entry points: keyframe re-projection blocks and their masked-L1 terms
    from nicer_slam_amd._native import WarpDesc
    assert lib.nsa_patch_warp_forward(None, 1, None, None, None, None, None) == NSA_EBADARG
    w = WarpDesc(2, 8, 40, 60, 4096, 4096, 4096, 4096, 4096, 4096, None, None)       # pointers never dereferenced
    assert lib.nsa_patch_warp_forward(ctypes.byref(w), 4, fake, fake, fake, fake, None) == NSA_EBADARG      # even patch size
    assert lib.nsa_patch_warp_forward(ctypes.byref(w), 5, fake, fake, fake, fake, None) == NSA_EBADARG      # patch > 1 without depth frames
    assert lib.nsa_patch_warp_backward(ctypes.byref(w), 1, fake, fake, fake, None, None, None) == NSA_EBADARG   # pose gradient without w2c gradient
    assert lib.nsa_patch_warp_backward(ctypes.byref(w), 5, fake, fake, None, None, None, None) == NSA_EBADARG   # patch > 1 needs the workspace
    assert lib.nsa_patch_warp_workspace(8, 1024, 1, 0) == 0
    assert lib.nsa_patch_warp_workspace(8, 1024, 5, 0) == 8 * 1024 * 25
    assert lib.nsa_flow_forward(ctypes.byref(w), None, None, 3, None, None) == NSA_EBADARG
    assert lib.nsa_flow_forward(ctypes.byref(w), None, None, 0, None, None) == 0                                # no edges: no-op
    assert lib.nsa_flow_backward(None, None, None, 0, None, None, None, None, None, None) == NSA_EBADARG
    assert lib.nsa_flow_workspace(8, 1024, 12, 0) == 0 and lib.nsa_flow_workspace(8, 1024, 12, 1) > 0
    assert lib.nsa_masked_l1(None, None, None, 16, 3, None, None, None, None) == NSA_EBADARG
    assert lib.nsa_masked_l1(fake, fake, None, 16, 0, fake, None, fake, None) == NSA_EBADARG                    # zero channels
    assert lib.nsa_masked_l1_workspace(1 << 20) >= 4
    # the folded tracking sequence (one launch each for begin / composite + L1 / ray reduction + tail)
    assert lib.nsa_track_begin(None, None, None, None, None, None, 8, None, None, None, None, None) == NSA_EBADARG
    assert lib.nsa_composite_track(fake, fake, fake, fake, fake, fake, 64, 8, 257, fake, 8, fake, fake, fake, fake, fake, None) == NSA_EBADARG   # > 256 samples per ray
    assert lib.nsa_composite_track(fake, fake, fake, fake, fake, fake, 64, 8, 128, fake, 4, fake, fake, fake, fake, fake, None) == NSA_EBADARG   # n_total < R
    assert lib.nsa_composite_track(None, None, None, None, None, None, 64, 0, 128, None, 0, None, None, None, None, None, None) == 0            # no rays
    assert lib.nsa_track_finish(fake, fake, fake, 8, 128, fake, fake, fake, fake, fake, 1, 0.0, None, None, None, 0.1, 0.9, 0.999, 1e-8, 0, 1.0,
                                None, fake, None) == NSA_EBADARG                                                                                 # Adam without its state
    assert lib.nsa_track_finish(fake, fake, fake, 8, 128, fake, fake, fake, fake, fake, 0, 8.0, None, None, None, 0.1, 0.9, 0.999, 1e-8, 0, 1.0,
                                fake, fake, None) == NSA_EBADARG                                                                                 # candidate without the step
    assert lib.nsa_track_finish_workspace(1024) == 256 * 16 + 4
    # empty work is a no-op, not an error (reference: a zero-size launch is never issued either)
    assert lib.nsa_sdf_points(None, 0, None, None, None, None, None, None) == 0
    assert lib.nsa_sampler_sdf(None, None, 0, 640, None, None, 0.0, 1.0, 3.5, None, None, None, None, None, None, None, None) == 0


def test_product_library_has_no_packed_fp32_arithmetic():
    """Build-time ISA rule (nicer_slam_amd/build.py::isa_check): v_pk_mul/add/fma_f32 -- what the SLP vectoriser makes of
    adjacent fp32 math -- must not appear in any gfx950 code object of the product library: with them the quad-tiling MFMA kernels
    are run-to-run irreproducible on MI355X (profiles/r04_slp_hazard_experiments.txt)."""
    import os
    import pytest
    from nicer_slam_amd import build
    if not os.path.exists(build.OBJDUMP):
        pytest.skip("llvm-objdump not available")
    assert build.isa_check(build.LIB) == {}
