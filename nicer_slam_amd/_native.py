"""ctypes binding of libnicer_slam_amd.so (the C ABI in include/nicer_slam_amd.h).

There is NO fallback: if the library is missing the import fails, so nothing above it can silently run
without the HIP kernels.
"""
import ctypes
import os

# PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7 -- the same
# SONAME as /opt/rocm's).  Whichever is loaded first serves the whole process, so torch must come first: our kernels
# then launch on the very runtime that owns torch's allocations and streams.  (Loading this library first made
# every launch fail on the GPU box.)
import torch  # noqa: F401  (must precede the CDLL below)

# NSA_LIB_TAG selects a side-by-side experiment build of the same library (nicer_slam_amd/build.py, NSA_BUILD_TAG);
# unset = the product library.  Either way it is the HIP library or nothing.
_TAG = os.environ.get("NSA_LIB_TAG", "")
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                         "libnicer_slam_amd" + ("_" + _TAG if _TAG else "") + ".so")

if not os.path.exists(_LIB_PATH):
    raise ImportError(
        f"{_LIB_PATH} not found: the HIP extension has not been built.  Run "
        "`python -m nicer_slam_amd.build` (needs hipcc; cross-compiles gfx950 without a GPU).")

lib = ctypes.CDLL(_LIB_PATH)

_p, _u32, _f32, _i = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_float, ctypes.c_int

lib.nsa_strerror.restype = ctypes.c_char_p
lib.nsa_strerror.argtypes = [_i]
lib.nsa_version.restype = _i
lib.nsa_hash_encode_forward.restype = _i
lib.nsa_hash_encode_forward.argtypes = [_p, _p, _p, _p, _u32, _u32, _u32, _u32, _f32, _u32, _i, _p, _p]
lib.nsa_hash_encode_backward.restype = _i
lib.nsa_hash_encode_backward.argtypes = [_p, _p, _p, _p, _p, _u32, _u32, _u32, _u32, _f32, _u32, _i, _p, _p, _p]
lib.nsa_hash_encode_second_backward.restype = _i
lib.nsa_hash_encode_second_backward.argtypes = [_p, _p, _p, _p, _u32, _u32, _u32, _u32, _f32, _u32, _i, _p, _p,
                                                _p, _p, _p]

EXPORTS = ["nsa_strerror", "nsa_version", "nsa_hash_encode_forward", "nsa_hash_encode_backward",
           "nsa_hash_encode_second_backward"]


def check(rc):
    if rc != 0:
        raise RuntimeError(lib.nsa_strerror(rc).decode())


class GridDesc(ctypes.Structure):
    """nsa_grid_t"""
    _fields_ = [("table", _p), ("offsets_host", _p), ("L", _u32), ("C", _u32), ("S", _f32), ("H", _u32),
                ("divide_factor", _f32), ("n_hidden", _u32), ("precision", _u32), ("tile", _u32)]


_gp = ctypes.POINTER(GridDesc)
lib.nsa_sampler_sdf.restype = _i
lib.nsa_sampler_sdf.argtypes = [_p, _p, _u32, _u32, _p, _p, _f32, _f32, _f32, _gp, _gp, _p, _p, _p, _p, _p, _p]
lib.nsa_sample_rays.restype = _i
lib.nsa_sample_rays.argtypes = [_p, _p, _p, _p, _p, _p, _u32, _u32, _u32, _u32, _p, _p, _u32, _f32, _p, _p, _p, _p]
EXPORTS += ["nsa_sampler_sdf", "nsa_sample_rays"]


class PointsDesc(ctypes.Structure):
    """nsa_points_t"""
    _fields_ = [("rays_o", _p), ("rays_d", _p), ("z_vals", _p), ("points", _p), ("P", _u32), ("S", _u32), ("order", _p)]


_pp = ctypes.POINTER(PointsDesc)
lib.nsa_sdfnet_forward.restype = _i
lib.nsa_sdfnet_forward.argtypes = [_pp, _gp, _p, _i, _p, _p, _p, _p]
lib.nsa_sdfnet_forward_pair.restype = _i
lib.nsa_sdfnet_forward_pair.argtypes = [_pp, _gp, _gp, _p, _p, _p, _p, _p, _p]
lib.nsa_sdfnet_backward.restype = _i
lib.nsa_sdfnet_backward.argtypes = [_pp, _gp, _p, _p, _p, _p, _i, _p, _p]
lib.nsa_colour_forward.restype = _i
lib.nsa_colour_forward.argtypes = [_pp, _gp, _p, _p, _p, _p, _p, _p]
lib.nsa_colour_backward.restype = _i
lib.nsa_colour_backward.argtypes = [_pp, _gp, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p]
lib.nsa_composite_forward.restype = _i
lib.nsa_composite_forward.argtypes = [_p, _p, _p, _p, _p, _p, _p, _u32, _u32, _u32, _p, _p, _p, _p, _p, _p]
lib.nsa_composite_backward.restype = _i
lib.nsa_composite_backward.argtypes = [_p, _p, _p, _p, _p, _p, _p, _u32, _u32, _u32, _p, _p, _p, _p, _p, _p, _p, _p, _p]
lib.nsa_rays_backward.restype = _i
lib.nsa_rays_backward.argtypes = [_p, _p, _p, _u32, _u32, _p, _p, _p]
lib.nsa_sdfnet_backward_params.restype = _i
lib.nsa_sdfnet_backward_params.argtypes = [_pp, _gp, _p, _p, _p, _p, _i, _p, _p, _p, _u32, _p]
lib.nsa_colour_forward_composite.restype = _i
lib.nsa_colour_forward_composite.argtypes = [_pp, _gp, _p, _p, _p, _p, _p, _p, _p, _u32, _p, _p, _p, _p, _p, _p]
EXPORTS += ["nsa_colour_forward_composite"]
lib.nsa_colour_forward_track.restype = _i
lib.nsa_colour_forward_track.argtypes = [_pp, _gp, _p, _p, _p, _p, _p, _p, _p, _u32, _p, _u32, _p, _p, _p, _p, _p, _p]
EXPORTS += ["nsa_colour_forward_track"]
lib.nsa_colour_coarse_backward.restype = _i
lib.nsa_colour_coarse_backward.argtypes = [_pp, _gp, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _gp, _p, _p, _p]
EXPORTS += ["nsa_colour_coarse_backward"]
lib.nsa_colour_backward_params.restype = _i
lib.nsa_colour_backward_params.argtypes = [_pp, _gp, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _u32, _p]
lib.nsa_sdfnet_emit_rows.restype = _i
lib.nsa_sdfnet_emit_rows.argtypes = []
lib.nsa_colour_emit_rows.restype = _i
lib.nsa_colour_emit_rows.argtypes = []
lib.nsa_sdfnet_emit_rows_nh.restype = _i
lib.nsa_sdfnet_emit_rows_nh.argtypes = [_u32]
lib.nsa_sdfnet_emit_rows_tile.restype = _i
lib.nsa_sdfnet_emit_rows_tile.argtypes = [_u32, _u32]
EXPORTS += ["nsa_sdfnet_backward_params", "nsa_colour_backward_params", "nsa_sdfnet_emit_rows", "nsa_colour_emit_rows",
            "nsa_sdfnet_emit_rows_nh", "nsa_sdfnet_emit_rows_tile"]
EXPORTS += ["nsa_sdfnet_forward", "nsa_sdfnet_forward_pair", "nsa_sdfnet_backward", "nsa_colour_forward", "nsa_colour_backward",
            "nsa_composite_forward", "nsa_composite_backward", "nsa_rays_backward"]

lib.nsa_rays_forward.restype = _i
lib.nsa_rays_forward.argtypes = [_p, _p, _p, _u32, _u32, _p, _p, _p, _p]
lib.nsa_rays_pose_backward.restype = _i
lib.nsa_rays_pose_backward.argtypes = [_p, _p, _p, _u32, _u32, _p, _p, _p, _p]
lib.nsa_rays_forward_draw.restype = _i
lib.nsa_rays_forward_draw.argtypes = [_p, _p, _p, _u32, _u32, _p, _p, _p, _p, ctypes.c_uint64, _p, _u32, _u32, _u32, _p, _p]
EXPORTS += ["nsa_rays_forward", "nsa_rays_forward_draw", "nsa_rays_pose_backward"]

lib.nsa_cam_to_pose.restype = _i
lib.nsa_cam_to_pose.argtypes = [_p, _u32, _p, _p]
lib.nsa_pose_grad_to_cam.restype = _i
lib.nsa_pose_grad_to_cam.argtypes = [_p, _p, _u32, _p, _p]
lib.nsa_l1_loss.restype = _i
lib.nsa_l1_loss.argtypes = [_p, _p, _u32, _p, _p, _p]
lib.nsa_adam_step.restype = _i
lib.nsa_adam_step.argtypes = [_p, _p, _p, _p, _p, _u32, _f32, _f32, _f32, _f32, _u32, _f32, _p]
EXPORTS += ["nsa_cam_to_pose", "nsa_pose_grad_to_cam", "nsa_l1_loss", "nsa_adam_step"]

lib.nsa_update_voxels.restype = _i
lib.nsa_update_voxels.argtypes = [_pp, _p, _u32, _p]
lib.nsa_adam_table_step.restype = _i
lib.nsa_adam_table_step.argtypes = [_p, _p, _p, _p, ctypes.c_uint64, _u32, _f32, _f32, _f32, _f32, _p]
lib.nsa_adam_table_step_clear.restype = _i
lib.nsa_adam_table_step_clear.argtypes = lib.nsa_adam_table_step.argtypes
lib.nsa_adam_table_step_zero_grad.restype = _i
lib.nsa_adam_table_step_zero_grad.argtypes = [_p, _p, _p, ctypes.c_uint64, _u32, _f32, _f32, _f32, _f32, _p]
EXPORTS += ["nsa_update_voxels", "nsa_adam_table_step", "nsa_adam_table_step_clear", "nsa_adam_table_step_zero_grad"]


class WnLayer(ctypes.Structure):
    """nsa_wn_layer_t"""
    _fields_ = [("weight_v", _p), ("weight_g", _p), ("bias", _p), ("rows", _u32), ("cols", _u32)]


lib.nsa_weight_norm_flat.restype = _i
lib.nsa_weight_norm_flat.argtypes = [ctypes.POINTER(WnLayer), _u32, _p, _p, _p]
lib.nsa_weight_norm_flat_backward.restype = _i
lib.nsa_weight_norm_flat_backward.argtypes = [ctypes.POINTER(WnLayer), _u32, _p, _p, _p, _p]
lib.nsa_emit_row.restype = _i
lib.nsa_emit_row.argtypes = [_p, _p, _p, _u32, ctypes.c_uint64, _f32, _p]
class AdamSeg(ctypes.Structure):
    """nsa_adam_seg_t"""
    _fields_ = [("param", _p), ("grad", _p), ("exp_avg", _p), ("exp_avg_sq", _p), ("n", _u32), ("step", _u32), ("lr", _f32)]


lib.nsa_adam_multi_step.restype = _i
lib.nsa_adam_multi_step.argtypes = [ctypes.POINTER(AdamSeg), _u32, _f32, _f32, _f32, _p]
EXPORTS += ["nsa_adam_multi_step"]
lib.nsa_fill_zero.restype = _i
lib.nsa_fill_zero.argtypes = [_p, ctypes.c_uint64, _p]
EXPORTS += ["nsa_weight_norm_flat", "nsa_weight_norm_flat_backward", "nsa_emit_row", "nsa_fill_zero"]

lib.nsa_emit_gemm.restype = _i
lib.nsa_emit_gemm.argtypes = [_p, ctypes.c_uint64, _u32, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), _u32,
                              _u32, ctypes.c_int, _p, _p, _p]
lib.nsa_emit_gemm_workspace.restype = ctypes.c_uint64
lib.nsa_emit_gemm_workspace.argtypes = [ctypes.c_uint64, _u32, _u32, ctypes.c_int]
EXPORTS += ["nsa_emit_gemm", "nsa_emit_gemm_workspace"]
lib.nsa_pack_blocks.restype = _i
lib.nsa_pack_blocks.argtypes = [_p, _p, ctypes.c_uint64, _p, ctypes.c_uint64, _p, ctypes.c_uint64, _p, _p]
lib.nsa_operand_form.restype = _i
lib.nsa_operand_form.argtypes = []
EXPORTS += ["nsa_pack_blocks", "nsa_operand_form"]


class FeedField(ctypes.Structure):
    """nsa_feed_field_t"""
    _fields_ = [("store", _p), ("out", _p), ("channels", _u32)]


class CopySeg(ctypes.Structure):
    """nsa_copy_seg_t"""
    _fields_ = [("dst", _p), ("src", _p), ("n", _u32)]


lib.nsa_copy_segments.restype = _i
lib.nsa_copy_segments.argtypes = [ctypes.POINTER(CopySeg), _u32, _p]
EXPORTS += ["nsa_copy_segments"]
lib.nsa_feed_gather.restype = _i
lib.nsa_feed_gather.argtypes = [ctypes.POINTER(FeedField), _u32, _p, _u32, _p, _u32, ctypes.c_uint64, _u32, _p, _p]
EXPORTS += ["nsa_feed_gather"]

class LossDesc(ctypes.Structure):
    """nsa_loss_t"""
    _fields_ = ([("bs", _u32), ("n", _u32), ("S", _u32), ("E", _u32)]
                + [(k, _p) for k in ("rgb", "rgb_gt", "depth", "depth_mono", "depth_real", "depth_real_mask", "mask_gt", "sdf",
                                     "normal", "normal_gt", "grad_theta", "grad_theta_nei")]
                + [(k, _f32) for k in ("w_rgb", "w_eik", "w_smooth", "w_depth", "w_gtdepth", "w_nl1", "w_ncos")]
                + [("depth_whole_image", ctypes.c_int)]
                + [(k, _p) for k in ("g_rgb", "g_depth", "g_normal", "g_theta", "g_theta_nei", "terms")])


lib.nsa_slam_loss.restype = _i
lib.nsa_slam_loss.argtypes = [ctypes.POINTER(LossDesc), _p, _p]
lib.nsa_slam_loss_workspace.restype = ctypes.c_uint64
lib.nsa_slam_loss_workspace.argtypes = [_u32, _u32, _u32]
EXPORTS += ["nsa_slam_loss", "nsa_slam_loss_workspace"]

lib.nsa_sdf_points.restype = _i
lib.nsa_sdf_points.argtypes = [_p, ctypes.c_uint64, _gp, _gp, _p, _p, _p, _p]
EXPORTS += ["nsa_sdf_points"]

lib.nsa_draw_picks.restype = _i
lib.nsa_draw_picks.argtypes = [_p, _u32, _u32, _u32, _u32, _p, _p, _p]
lib.nsa_draw.restype = _i
lib.nsa_draw.argtypes = [_p, ctypes.c_uint64, _p, _u32, _u32, _u32, _u32, _p, _p, _p]
EXPORTS += ["nsa_draw_picks", "nsa_draw"]

lib.nsa_track_head.restype = _i
lib.nsa_track_head.argtypes = [_p, _p, _p, _u32, _p, _p, _p, _p, _p]
lib.nsa_track_tail.restype = _i
lib.nsa_track_tail.argtypes = [_p, _p, _p, _u32, _p, _p, _p, _i, _f32, _p, _p, _p, _f32, _f32, _f32, _f32, _u32, _f32, _p, _p,
                               _p]
EXPORTS += ["nsa_track_head", "nsa_track_tail"]
lib.nsa_track_begin.restype = _i
lib.nsa_track_begin.argtypes = [_p, _p, _p, _p, _p, _p, _u32, _p, _p, _p, _p, _p]
lib.nsa_composite_track.restype = _i
lib.nsa_composite_track.argtypes = [_p, _p, _p, _p, _p, _p, _u32, _u32, _u32, _p, _u32, _p, _p, _p, _p, _p, _p]
lib.nsa_track_finish.restype = _i
lib.nsa_track_finish.argtypes = [_p, _p, _p, _u32, _u32, _p, _p, _p, _p, _p, _i, _f32, _p, _p, _p, _f32, _f32, _f32, _f32, _u32,
                                 _f32, _p, _p, _p]
lib.nsa_track_finish_workspace.restype = ctypes.c_uint64
lib.nsa_track_finish_workspace.argtypes = [_u32]
lib.nsa_track_begin_draw.restype = _i
lib.nsa_track_begin_draw.argtypes = [_p, _p, _p, _p, _p, _p, _u32, _p, _p, _p, _p, _p, ctypes.c_uint64, _p, _u32, _u32, _u32, _p, _p]
EXPORTS += ["nsa_track_begin", "nsa_track_begin_draw", "nsa_composite_track", "nsa_track_finish", "nsa_track_finish_workspace"]

lib.nsa_morton_keys.restype = _i
lib.nsa_morton_keys.argtypes = [_pp, _p, _p]
lib.nsa_morton_order.restype = _i
lib.nsa_morton_order.argtypes = [_pp, _p, _p, _u32, _p]
lib.nsa_morton_order_workspace.restype = ctypes.c_uint64
lib.nsa_morton_order_workspace.argtypes = [_u32]
EXPORTS += ["nsa_morton_order", "nsa_morton_order_workspace"]
EXPORTS += ["nsa_morton_keys"]

lib.nsa_adam_step_scaled.restype = _i
lib.nsa_adam_step_scaled.argtypes = [_p, _p, _p, _p, _p, _p, _u32, _f32, _f32, _f32, _f32, _u32, _f32, _p, _p, _p]
EXPORTS += ["nsa_adam_step_scaled"]


class WarpDesc(ctypes.Structure):
    """nsa_warp_t"""
    _fields_ = ([("b", _u32), ("n", _u32), ("H", _u32), ("W", _u32)]
                + [(k, _p) for k in ("uv", "pose", "w2c", "K", "depth", "images", "depths", "frame_index")])


_wp, _u64 = ctypes.POINTER(WarpDesc), ctypes.c_uint64
lib.nsa_patch_warp_forward.restype = _i
lib.nsa_patch_warp_forward.argtypes = [_wp, _u32, _p, _p, _p, _p, _p]
lib.nsa_patch_warp_backward.restype = _i
lib.nsa_patch_warp_backward.argtypes = [_wp, _u32, _p, _p, _p, _p, _p, _p]
lib.nsa_patch_warp_workspace.restype = _u64
lib.nsa_patch_warp_workspace.argtypes = [_u32, _u32, _u32, _i]
lib.nsa_flow_forward.restype = _i
lib.nsa_flow_forward.argtypes = [_wp, _p, _p, _u32, _p, _p]
lib.nsa_flow_backward.restype = _i
lib.nsa_flow_backward.argtypes = [_wp, _p, _p, _u32, _p, _p, _p, _p, _p, _p]
lib.nsa_flow_workspace.restype = _u64
lib.nsa_flow_workspace.argtypes = [_u32, _u32, _u32, _i]
lib.nsa_masked_l1.restype = _i
lib.nsa_masked_l1.argtypes = [_p, _p, _p, _u64, _u32, _p, _p, _p, _p]
lib.nsa_masked_l1_workspace.restype = _u64
lib.nsa_masked_l1_workspace.argtypes = [_u64]
EXPORTS += ["nsa_patch_warp_forward", "nsa_patch_warp_backward", "nsa_patch_warp_workspace", "nsa_flow_forward",
            "nsa_flow_backward", "nsa_flow_workspace", "nsa_masked_l1", "nsa_masked_l1_workspace"]
