"""Function-level goldens captured from the reference (tests/golden/make_golden.py::func_case) for the rows that are plain
PyTorch there -- a1 rays, a2 cube intersection, a7 positional encoding, a11 density, a13 compositing weights, a17 pose
parametrisation -- against (i) the oracle's restatement and (ii) the product's host-side twins, on CPU."""
import numpy as np
import pytest
import torch

from helpers import load, tt, assert_close


@pytest.fixture(scope="module")
def fx():
    return load("func_rows")


def test_a17_pose_parametrisation(fx):
    from oracle import render_ref as R
    from nicer_slam_amd.utils import general
    cam = tt(fx["a17_cam"])
    assert_close(R.camera_from_tensor(cam), fx["a17_pose"], 1e-6, 1e-6, "oracle camera_from_tensor")
    assert_close(R.quad2rotation(cam[:, :4]), fx["a17_rot"], 1e-6, 1e-6, "oracle quad2rotation")
    assert_close(general.get_camera_from_tensor(cam), fx["a17_pose"], 1e-6, 1e-6, "product get_camera_from_tensor")
    assert_close(general.quad2rotation(cam[:, :4]), fx["a17_rot"], 1e-6, 1e-6, "product quad2rotation")


def test_a1_rays(fx):
    from oracle import render_ref as R
    from nicer_slam_amd.utils import rend_util
    uv, K, pose = tt(fx["a1_uv"]), tt(fx["a1_K"]), tt(fx["a1_pose"])
    d, o = rend_util.get_camera_params(uv, pose, K)
    assert_close(d, fx["a1_ray_dirs"], 1e-6, 1e-5, "product ray_dirs")
    assert_close(o, fx["a1_cam_loc"], 0, 0, "product cam_loc")
    out = R.camera_rays(uv, pose, K)
    assert_close(out[0].reshape(fx["a1_ray_dirs"].shape), fx["a1_ray_dirs"], 1e-6, 1e-5, "oracle ray_dirs")


def test_a2_cube_intersection(fx):
    from oracle import render_ref as R
    from nicer_slam_amd.model.ray_sampler import UniformSampler
    o, d = tt(fx["a2_o"]), tt(fx["a2_d"])
    us = UniformSampler(1.0, 0.0, 16, far=3.5)
    near, far = us.near_far_from_cube(o.clone(), d.clone(), 1.0)
    assert_close(near, fx["a2_near"], 1e-6, 1e-6, "product near")
    assert_close(far, fx["a2_far"], 1e-6, 1e-6, "product far")
    assert_close(R.cube_far(o, d, 1.0, 3.5).reshape(-1, 1), fx["a2_far"], 1e-6, 1e-6, "oracle far")
    assert float(fx["a2_far"].min()) < 0 < float(fx["a2_far"].max())   # incl. a ray whose cube lies behind its origin


def test_a7_positional_encoding(fx):
    from oracle import render_ref as R
    from nicer_slam_amd.model.embedder import get_embedder
    x = tt(fx["a7_x"])
    for m in (6, 4):
        fn, dim = get_embedder(m, input_dims=3)
        assert dim == fx[f"a7_pe{m}"].shape[1]
        assert_close(fn(x), fx[f"a7_pe{m}"], 1e-6, 1e-6, f"product PE{m}")
        assert_close(R.positional_encoding(x, m), fx[f"a7_pe{m}"], 1e-6, 1e-6, f"oracle PE{m}")


def test_a11_density(fx):
    from oracle import render_ref as R
    from nicer_slam_amd.model.density import GridPredefineDensity
    vox, x, sdf = tt(fx["a11_voxels"]), tt(fx["a11_x"]), tt(fx["a11_sdf"])
    assert bool((x.abs() > 0.99).any(1).any()) and not bool((x.abs() > 0.99).any(1).all())
    dens = GridPredefineDensity()
    dens.voxels, dens.voxel_res = vox, 64
    assert_close(dens.get_beta(x), fx["a11_beta"], 1e-8, 1e-6, "product beta")
    assert_close(dens(sdf, x=x), fx["a11_sigma"], 1e-5, 1e-5, "product sigma")
    assert_close(R.beta_from_voxels(vox, x, 64).reshape(-1, 1), fx["a11_beta"], 1e-8, 1e-6, "oracle beta")
    assert_close(R.density(sdf, x, vox, 64).reshape(-1, 1), fx["a11_sigma"], 1e-5, 1e-5, "oracle sigma")


def test_a13_compositing_weights(fx):
    from oracle import render_ref as R
    from nicer_slam_amd.model.density import GridPredefineDensity
    from nicer_slam_amd.model.ray_sampler import transmittance_weights
    vox = tt(fx["a11_voxels"])
    z, sdf, x = tt(fx["a13_z"]), tt(fx["a13_sdf"]), tt(fx["a13_x"])
    assert_close(R.volume_weights(z, sdf, x, vox, 64), fx["a13_weights"], 1e-6, 1e-5, "oracle weights")
    dens = GridPredefineDensity()
    dens.voxels, dens.voxel_res = vox, 64
    w = transmittance_weights(z, dens(sdf, x=x).reshape(-1, z.shape[1]))
    assert_close(w, fx["a13_weights"], 1e-6, 1e-5, "product weights")
    assert float(np.abs(fx["a13_weights"]).sum()) > 0
