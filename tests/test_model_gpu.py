"""The model plug-in (SLAMNetwork) on the GPU vs the goldens captured from the reference: forward dict, pose
gradient and every parameter gradient, tracking and mapping, both stages."""
import numpy as np
import pytest
import torch

from helpers import load, tt, draws_of, golden_objective, assert_close, assert_outputs_close
from test_model_cpu import build_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("engine", ["composed"])
@pytest.mark.parametrize("name", ["full_tracking", "full_tracking_poisson", "full_mapping", "full_mapping_coarse_base",
                                  "full_vis_eval", "full_tracking_rw", "full_mapping_rw", "full_mapping_rw_coarse",
                                  "full_tracking_7scenes", "full_mapping_7scenes", "full_mapping_7scenes_coarse_base"])
def test_forward_and_grads_vs_reference_goldens(name, engine):
    from nicer_slam_amd.utils.general import camera_from_tensor_torch as get_camera_from_tensor   # the reference's op order: golden comparison (on-face far samples)
    fx = load(name)
    model = build_model(fx).cuda()
    model.engine = engine
    mode, stage, cstage = str(fx["meta_mode"]), str(fx["meta_stage"]), str(fx["meta_color_stage"])
    model.train(bool(fx["meta_training"]))
    model.voxels = tt(fx["in_voxels"]).cuda()
    model.draws = draws_of(fx, "cuda")
    model.draws["z_vals_override"] = tt(fx["out_z_vals"]).cuda()
    cam = tt(fx["in_cam"]).cuda().requires_grad_(True)
    pose = get_camera_from_tensor(cam)
    out = model({"intrinsics": tt(fx["in_K"]).cuda(), "uv": tt(fx["in_uv"]).cuda(), "pose": pose},
                torch.arange(pose.shape[0], device="cuda"), {}, mode=mode, stage=stage, color_stage=cstage, frame_idx=1)
    assert_outputs_close(out, fx, ("depth_vals", "sdf", "weights", "rgb", "rgb_values", "depth_values", "entropy", "normal_map",
                                   "grad_theta", "grad_theta_nei"))
    assert_close(model.voxels, fx["out_voxels"], 0, 0, "voxels")
    if not model.training:
        return
    loss = golden_objective(out, fx, mode)
    loss.backward()
    assert_close(cam.grad, fx["grad_cam"], 2e-6, 1e-3, "grad_cam")
    for n, p in model.named_parameters():
        ref = fx["grad_" + n]
        if ref.size == 0:
            assert p.grad is None or float(p.grad.abs().max()) == 0
        else:
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            assert_close(g, ref, 1e-6 + 2e-4 * float(np.abs(ref).max()), 1e-3, "grad " + n)


def test_patch_warp_block_on_gpu():
    """Mapping mode incl. the patch-warp gather (composed engine on the GPU) vs the reference golden."""
    from nicer_slam_amd.utils.general import camera_from_tensor_torch as get_camera_from_tensor   # the reference's op order: golden comparison (on-face far samples)
    fx = load("full_mapping_warp")
    model = build_model(fx).cuda()
    model.train(True)
    model.voxels = tt(fx["in_voxels"]).cuda()
    model.draws = draws_of(fx, "cuda")
    model.draws["z_vals_override"] = tt(fx["out_z_vals"]).cuda()
    cam = tt(fx["in_cam"]).cuda().requires_grad_(True)
    out = model({"intrinsics": tt(fx["in_K"]).cuda(), "uv": tt(fx["in_uv"]).cuda(), "pose": get_camera_from_tensor(cam)},
                torch.arange(2, device="cuda"),
                {"full_rgb": tt(fx["in_full_rgb"]).cuda(), "full_depth": tt(fx["in_full_depth"]).cuda()},
                mode="mapping", stage="fine", color_stage="highfreq", frame_idx=1)
    loss = (out["rgb_values"].reshape(-1, 3) - tt(fx["gt_rgb"]).cuda()).abs().mean()
    for ps, (gt_w, samp, mask, ray_mask) in out["warp_output"].items():
        assert_close(gt_w, fx[f"out_warp{ps}_gt"], 0, 0, f"gt patch {ps}")
        assert float((mask.cpu() != tt(fx[f"out_warp{ps}_mask"])).float().mean()) < 0.01
        assert_close(samp * mask[..., None], tt(fx[f"out_warp{ps}_sampled"]) * tt(fx[f"out_warp{ps}_mask"])[..., None].float(),
                     5e-5, 1e-3, f"sampled {ps}")
        loss = loss + 0.5 * ((gt_w - samp).abs().sum(-1) * mask.float()).sum() / (mask.float().sum() + 1)
    loss.backward()
    assert_close(loss, fx["out_loss"], 1e-4, 1e-4, "loss")
    assert_close(cam.grad, fx["grad_cam"], 2e-3 * float(np.abs(fx["grad_cam"]).max()), 2e-3, "grad_cam")
