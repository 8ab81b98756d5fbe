"""The per-row kernels against the function-level goldens of the reference (tests/golden/func_rows.npz): a17 cam->pose,
a1 pixel->ray lifting (skewed intrinsics), a2 cube exit distance, a11/a13 density + compositing weights."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import load, tt, assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx():
    return load("func_rows")


def _st():
    return torch.cuda.current_stream().cuda_stream


def test_a17_cam_to_pose_kernel(fx):
    from nicer_slam_amd._native import lib, check
    cam = tt(fx["a17_cam"]).cuda().contiguous()
    pose = torch.empty(3, 4, 4, device="cuda")
    check(lib.nsa_cam_to_pose(cam.data_ptr(), 3, pose.data_ptr(), _st()))
    assert_close(pose, fx["a17_pose"], 1e-6, 1e-6, "pose")


def test_a17_get_camera_from_tensor_on_the_device_is_the_kernel_pair(fx):
    """nicer_slam_amd.utils.general.get_camera_from_tensor on a device tensor = one HIP kernel forward, one backward: the value
    against the reference golden, the gradient against torch autograd through the restated quad2rotation (non-unit quaternions)."""
    from nicer_slam_amd.utils import general as G
    cam = tt(fx["a17_cam"]).cuda().contiguous().requires_grad_(True)
    pose = G.get_camera_from_tensor(cam)
    assert pose.grad_fn is not None and "CamToPose" in type(pose.grad_fn).__name__
    assert_close(pose, fx["a17_pose"], 1e-6, 1e-6, "pose")
    g = torch.randn(pose.shape, generator=torch.Generator().manual_seed(0)).cuda()
    (pose * g).sum().backward()
    cam_c = tt(fx["a17_cam"]).clone().requires_grad_(True)
    (G.get_camera_from_tensor(cam_c) * g.cpu()).sum().backward()          # CPU tensors: the torch restatement
    assert_close(cam.grad, cam_c.grad, 1e-6, 1e-5, "d/d cam")
    one = G.get_camera_from_tensor(cam.detach()[0])                       # single 7-vector -> [4,4]
    assert one.shape == (4, 4)
    assert_close(one, fx["a17_pose"][0], 1e-6, 1e-6, "single pose")


def test_a1_rays_kernel(fx):
    from nicer_slam_amd._native import lib, check
    uv, K, pose = tt(fx["a1_uv"]).cuda().contiguous(), tt(fx["a1_K"]).cuda().contiguous(), tt(fx["a1_pose"]).cuda().contiguous()
    b, n, _ = uv.shape
    o = torch.empty(b * n, 3, device="cuda")
    d = torch.empty(b * n, 3, device="cuda")
    ds = torch.empty(b * n, device="cuda")
    check(lib.nsa_rays_forward(uv.data_ptr(), pose.data_ptr(), K.data_ptr(), b, n, o.data_ptr(), d.data_ptr(), ds.data_ptr(), _st()))
    assert_close(d.view(b, n, 3), fx["a1_ray_dirs"], 1e-6, 1e-5, "ray_dirs")
    assert_close(o.view(b, n, 3)[:, 0], fx["a1_cam_loc"], 0, 0, "cam_loc")
    # single-image fused head (tracker): same rays for image 0
    cam = tt(fx["a17_cam"])[:1].cuda().contiguous()
    pose1 = torch.empty(1, 4, 4, device="cuda")
    o1, d1, ds1 = torch.empty(n, 3, device="cuda"), torch.empty(n, 3, device="cuda"), torch.empty(n, device="cuda")
    check(lib.nsa_track_head(uv[0].contiguous().data_ptr(), K[0].contiguous().data_ptr(), cam.data_ptr(), n, pose1.data_ptr(),
                             o1.data_ptr(), d1.data_ptr(), ds1.data_ptr(), _st()))
    assert_close(d1, fx["a1_ray_dirs"][0], 1e-6, 1e-5, "ray_dirs (track head)")
    assert_close(pose1[0], fx["a17_pose"][0], 1e-6, 1e-6, "pose (track head)")


def test_a11_a13_density_and_weights_in_the_composite_kernel(fx):
    """nsa_composite_forward (density from the visit counter at x = o + z d, Laplace density, compositing weights, ray
    sums) on straight rays built from the golden's z and sdf values, against the oracle's volume_weights -- which
    test_func_rows_cpu.py pins to the reference's own SLAMNetwork.volume_rendering / GridPredefineDensity outputs."""
    from oracle import render_ref as R
    from nicer_slam_amd._native import lib, check
    vox = tt(fx["a11_voxels"])
    z = tt(fx["a13_z"])
    Rn, S = z.shape
    g = torch.Generator().manual_seed(2)
    o = (torch.rand(Rn, 3, generator=g) - 0.5) * 0.4
    d = torch.nn.functional.normalize(torch.rand(Rn, 3, generator=g) - 0.5, dim=-1) * 0.45
    sdf = tt(fx["a13_sdf"]).reshape(Rn, S)
    x = (o[:, None] + z[..., None] * d[:, None]).reshape(-1, 3)
    ref_w = R.volume_weights(z, sdf.reshape(-1, 1), x, vox, 64)
    dev = lambda t: t.cuda().contiguous()
    oz, dz, zz, sd, vx = dev(o), dev(d), dev(z), dev(sdf.reshape(-1)), dev(vox)
    rgb = torch.rand(Rn * S, 3, device="cuda")
    grad = torch.randn(Rn * S, 3, device="cuda")
    w = torch.empty(Rn, S, device="cuda")
    rgbv, dep, nm, ent = (torch.empty(Rn, 3, device="cuda"), torch.empty(Rn, device="cuda"), torch.empty(Rn, 3, device="cuda"),
                          torch.empty(Rn, device="cuda"))
    check(lib.nsa_composite_forward(oz.data_ptr(), dz.data_ptr(), zz.data_ptr(), sd.data_ptr(), rgb.data_ptr(), grad.data_ptr(),
                                    vx.data_ptr(), 64, Rn, S, w.data_ptr(), rgbv.data_ptr(), dep.data_ptr(), nm.data_ptr(),
                                    ent.data_ptr(), _st()))
    assert_close(w, ref_w.numpy(), 1e-6, 1e-5, "weights")
    assert_close(rgbv, (ref_w[..., None] * rgb.cpu().view(Rn, S, 3)).sum(1).numpy(), 2e-6, 1e-5, "rgb_values")
    assert_close(dep, ((ref_w * z).sum(1) / (ref_w.sum(1) + 1e-8)).numpy(), 2e-6, 1e-5, "depth")
