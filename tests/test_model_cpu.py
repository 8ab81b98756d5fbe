"""Host logic of the product's model layer (nicer_slam_amd/model, composed engine) against the goldens captured from
the reference -- on CPU, with the oracle injected at the native seam (``hashgrid._backend``), i.e. everything
except the HIP kernels themselves: module structure, state_dict key compatibility, sampler, composite, RNG plumbing."""
import numpy as np
import pytest
import torch

from helpers import load, tt, params_of, draws_of, golden_objective, assert_close


@pytest.fixture()
def oracle_seam(monkeypatch):
    from oracle import hashenc
    from nicer_slam_amd.hashencoder import hashgrid
    monkeypatch.setattr(hashgrid, "_backend", hashenc.OracleBackend())
    return hashgrid


def build_model(fx):
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import model_conf
    cg, fg, col = fx["meta_coarse_grid"], fx["meta_fine_grid"], fx["meta_colour_grid"]
    ns, ne, nx = [int(v) for v in fx["meta_samples"]]
    warp = "meta_img_res" in fx
    family = str(fx["meta_family"]) if "meta_family" in fx else "replica"      # 7-Scenes / Azure model subtree (utils/conf.py)
    conf = model_conf(family, ns, ne, nx, use_warp_loss=warp)
    if warp:
        conf["mapping_patchsizes"] = [1, 5]
    for net, g in (("coarse", cg), ("fine", fg)):
        conf["implicit_network"][net].update(base_size=int(g[0]), end_size=int(g[1]), logmap=int(g[2]),
                                             num_levels=int(g[3]), level_dim=int(g[4]))

    class DS:
        img_res = tuple(int(v) for v in fx["meta_img_res" if warp else "meta_frame_res"]) if (warp or "meta_frame_res" in fx) \
            else (680, 1200)
    model = SLAMNetwork(conf, dataset=DS(), n_images=4,
                        colour_grid=dict(base_resolution=int(col[0]), desired_resolution=int(col[1]),
                                         log2_hashmap_size=int(col[2])))
    missing, unexpected = model.load_state_dict(params_of(fx), strict=True), None
    return model


@pytest.mark.parametrize("name", ["full_tracking", "full_mapping", "full_mapping_coarse_base", "full_vis_eval",
                                  "full_tracking_rw", "full_mapping_rw", "full_tracking_7scenes", "full_mapping_7scenes_coarse_base"])
def test_model_layer_matches_reference(oracle_seam, name):
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    fx = load(name)
    model = build_model(fx)
    # state_dict names are the reference's (checkpoint compatibility, volsdf_train.py:226-253)
    assert set(model.state_dict().keys()) == {k[len("param_"):] for k in fx if k.startswith("param_")}
    mode, stage, cstage = str(fx["meta_mode"]), str(fx["meta_stage"]), str(fx["meta_color_stage"])
    model.train(bool(fx["meta_training"]))
    model.voxels = tt(fx["in_voxels"]).clone()
    assert model.density.voxels is model.voxels
    model.draws = draws_of(fx)
    model.draws["z_vals_override"] = tt(fx["out_z_vals"])
    cam = tt(fx["in_cam"]).requires_grad_(True)
    pose = get_camera_from_tensor(cam)
    assert_close(pose, fx["in_pose"], 1e-7, 1e-6, "pose")
    out = model({"intrinsics": tt(fx["in_K"]), "uv": tt(fx["in_uv"]), "pose": pose}, torch.arange(pose.shape[0]), {},
                mode=mode, stage=stage, color_stage=cstage, frame_idx=1)
    for k in ("depth_vals", "sdf", "weights", "rgb", "rgb_values", "depth_values", "entropy", "normal_map",
              "grad_theta", "grad_theta_nei"):
        if "out_" + k in fx:
            assert_close(out[k], fx["out_" + k], 1e-5, 1e-4, k)
    assert_close(model.voxels, fx["out_voxels"], 0, 0, "voxels")
    if not model.training:
        return
    loss = golden_objective(out, fx, mode)
    loss.backward()
    assert_close(cam.grad, fx["grad_cam"], 1e-6, 1e-3, "grad_cam")
    for n, p in model.named_parameters():
        ref = fx["grad_" + n]
        if ref.size == 0:
            assert p.grad is None or float(p.grad.abs().max()) == 0
        else:
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            assert_close(g, ref, 1e-6 + 1e-4 * float(np.abs(ref).max()), 1e-3, "grad " + n)


def test_patch_warp_block_matches_reference(oracle_seam):
    """Mapping mode with use_warp_loss (all shipped configs enable it): the warp gather's four outputs per patch size
    and the pose gradient of an objective that runs through it (reference network.py:167-279)."""
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    fx = load("full_mapping_warp")
    model = build_model(fx)
    model.train(True)
    model.voxels = tt(fx["in_voxels"]).clone()
    model.draws = draws_of(fx)
    model.draws["z_vals_override"] = tt(fx["out_z_vals"])
    cam = tt(fx["in_cam"]).requires_grad_(True)
    out = model({"intrinsics": tt(fx["in_K"]), "uv": tt(fx["in_uv"]), "pose": get_camera_from_tensor(cam)},
                torch.arange(2), {"full_rgb": tt(fx["in_full_rgb"]), "full_depth": tt(fx["in_full_depth"])},
                mode="mapping", stage="fine", color_stage="highfreq", frame_idx=1)
    assert_close(out["rgb_values"], fx["out_rgb_values"], 1e-5, 1e-4, "rgb_values")
    loss = (out["rgb_values"].reshape(-1, 3) - tt(fx["gt_rgb"])).abs().mean()
    assert sorted(out["warp_output"]) == [1, 5]
    for ps, (gt_w, samp, mask, ray_mask) in out["warp_output"].items():
        assert_close(gt_w, fx[f"out_warp{ps}_gt"], 0, 0, f"gt patch {ps}")
        assert bool((mask == tt(fx[f"out_warp{ps}_mask"])).all()), f"mask {ps}"
        assert_close(samp, fx[f"out_warp{ps}_sampled"], 2e-5, 1e-4, f"sampled {ps}")
        if ps > 1:
            assert bool((ray_mask == tt(fx[f"out_warp{ps}_raymask"])).all())
        else:
            assert ray_mask is None
        loss = loss + 0.5 * ((gt_w - samp).abs().sum(-1) * mask.float()).sum() / (mask.float().sum() + 1)
    assert_close(loss, fx["out_loss"], 1e-6, 1e-5, "loss")
    loss.backward()
    assert_close(cam.grad, fx["grad_cam"], 1e-4 * float(np.abs(fx["grad_cam"]).max()), 1e-3, "grad_cam")


def test_sampler_free_running(oracle_seam):
    """Sampler of the model layer without the z override: compare in CDF space (see test_oracle_golden)."""
    from oracle import render_ref as R
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    fx = load("full_tracking")
    model = build_model(fx)
    model.train(True)
    model.draws = draws_of(fx)
    pose = get_camera_from_tensor(tt(fx["in_cam"]))
    with torch.no_grad():
        from nicer_slam_amd.utils import rend_util
        d, o = rend_util.get_camera_params(tt(fx["in_uv"]), pose, tt(fx["in_K"]))
        assert_close(d, fx["out_ray_dirs"], 1e-9, 1e-6, "dirs")
        z, _ = model.ray_sampler.get_z_vals(d.reshape(-1, 3), o.repeat(d.shape[1], 1), model)
    zr = tt(fx["out_z_vals"])
    assert float(((z - zr).abs() <= 1e-5 + 1e-4 * zr.abs()).float().mean()) >= 0.97
    assert bool((z[:, 1:] >= z[:, :-1]).all())


def test_get_tensor_from_camera_roundtrip():
    from nicer_slam_amd.utils.general import get_camera_from_tensor, get_tensor_from_camera
    g = torch.Generator().manual_seed(0)
    for _ in range(20):
        q = torch.randn(4, generator=g)
        q = q / q.norm()
        if q[0] < 0:
            q = -q
        cam = torch.cat([q, torch.randn(3, generator=g)])
        back = get_tensor_from_camera(get_camera_from_tensor(cam))
        assert_close(back, cam, 1e-5, 1e-5, "roundtrip")


def test_get_tensor_from_camera_known_answers():
    """Hand-derived known answers for general.py:103-126 (mathutils ``Matrix(R).to_quaternion()``: Hamilton quaternion (w, x, y, z)
    of the ACTIVE rotation R, the same convention quad2rotation (general.py:52-76) inverts, w >= 0 representative):
      rotation by angle a about unit axis n  <->  q = (cos a/2, n sin a/2);
      R_z(90 deg)  = [[0,-1,0],[1,0,0],[0,0,1]]        -> (sqrt(1/2), 0, 0, sqrt(1/2))
      120 deg about (1,1,1)/sqrt(3): cyclic permutation [[0,0,1],[1,0,0],[0,1,0]]  (x->y->z->x)  -> (1/2, 1/2, 1/2, 1/2)
      R_x(180 deg) = diag(1,-1,-1)                     -> (0, +-1, 0, 0)        (w = 0: the sign is free, q and -q are one rotation)
      R_y(90 deg)  = [[0,0,1],[0,1,0],[-1,0,0]]        -> (sqrt(1/2), 0, sqrt(1/2), 0)
      178 deg about (0, 0.6, 0.8)                       -> (cos 89, 0, 0.6 sin 89, 0.8 sin 89)  (trace < 0 branch)
    and the layout: [q, T] by default, [T, q] with Tquad=True; a 3x4 matrix is accepted like a 4x4."""
    import math
    from nicer_slam_amd.utils.general import get_tensor_from_camera, quad2rotation
    r = math.sqrt(0.5)
    a = math.radians(178.0)
    n = (0.0, 0.6, 0.8)
    c, s_, v = math.cos(a), math.sin(a), 1 - math.cos(a)
    R178 = [[c + n[0] * n[0] * v, n[0] * n[1] * v - n[2] * s_, n[0] * n[2] * v + n[1] * s_],          # Rodrigues' formula
            [n[1] * n[0] * v + n[2] * s_, c + n[1] * n[1] * v, n[1] * n[2] * v - n[0] * s_],
            [n[2] * n[0] * v - n[1] * s_, n[2] * n[1] * v + n[0] * s_, c + n[2] * n[2] * v]]
    cases = [([[0, -1, 0], [1, 0, 0], [0, 0, 1]], (r, 0, 0, r)),
             ([[0, 0, 1], [1, 0, 0], [0, 1, 0]], (0.5, 0.5, 0.5, 0.5)),
             ([[1, 0, 0], [0, -1, 0], [0, 0, -1]], (0, 1, 0, 0)),
             ([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], (r, 0, r, 0)),
             (R178, (math.cos(a / 2), 0.0, 0.6 * math.sin(a / 2), 0.8 * math.sin(a / 2))),
             ([[1, 0, 0], [0, 1, 0], [0, 0, 1]], (1, 0, 0, 0))]
    T = torch.tensor([0.3, -1.25, 2.0])
    for R, q in cases:
        RT = torch.eye(4, dtype=torch.float64)
        RT[:3, :3] = torch.tensor(R, dtype=torch.float64)
        RT[:3, 3] = T.double()
        got = get_tensor_from_camera(RT)
        want = torch.tensor(q, dtype=torch.float32)
        assert got.shape == (7,) and got.dtype == torch.float32
        if abs(q[0]) < 1e-9:                       # half-turn: either sign
            assert min(float((got[:4] - want).abs().max()), float((got[:4] + want).abs().max())) < 1e-6, (R, got)
        else:
            assert float(got[0]) > 0
            assert_close(got[:4], want.numpy(), 1e-6, 1e-6, f"quaternion of {R}")
        assert_close(got[4:], T.numpy(), 0, 0, "translation")
        assert_close(quad2rotation(got[None, :4].double())[0], RT[:3, :3].numpy(), 1e-6, 1e-6, "quad2rotation inverts it")
        tq = get_tensor_from_camera(RT, Tquad=True)
        assert torch.equal(tq[:3], got[4:]) and torch.equal(tq[3:], got[:4])
        assert torch.equal(get_tensor_from_camera(RT[:3]), got)          # 3x4


def test_backend_rejects_cpu_tensors():
    """No CPU fallback: the native seam refuses host tensors like the reference's CHECK_CUDA (hashencoder.cu:16)."""
    from nicer_slam_amd.hashencoder.hashgrid import HashEncoder
    enc = HashEncoder(num_levels=2, level_dim=2, base_resolution=4, desired_resolution=8, log2_hashmap_size=8)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        enc(torch.zeros(4, 3))


def test_mapping_host_guards_without_a_gpu():
    """Host-side refusals of the round-5 mapping helpers: the in-place table-gradient buffers are for CUDA float32 tables only (no
    CPU path to fall into), and a clearing policy that no longer exists is named, not silently mapped."""
    import os
    import pytest
    import torch
    from nicer_slam_amd.fused import tablegrad
    with pytest.raises(RuntimeError, match="CUDA"):
        tablegrad.target(torch.nn.Parameter(torch.zeros(8, 2)))
    assert tablegrad.consumable(torch.nn.Parameter(torch.zeros(2)), torch.zeros(2)) is False
    from nicer_slam_amd.optim import Adam
    assert Adam([torch.nn.Parameter(torch.zeros(3))]).consume_table_grads is False       # the clearing policy is a constructor argument
    assert Adam([torch.nn.Parameter(torch.zeros(3))], consume_table_grads=True).consume_table_grads is True
    with pytest.raises(ValueError, match="none_grad"):
        Adam([torch.nn.Parameter(torch.zeros(3))], none_grad="async")
    with pytest.raises(RuntimeError):                    # the optimizer itself: no CPU fallback either (no GPU / not a CUDA tensor)
        p = torch.nn.Parameter(torch.zeros(3))
        p.grad = torch.ones(3)
        Adam([p]).step()


def _parse_conf(text):
    """the subset of HOCON the shipped run configs use: `name { ... }`, `key = value`, `key = [` ... `]` over several lines"""
    root, stack, lst = {}, [], None
    cur = root
    for raw in text.splitlines():
        line = raw.strip()
        if not line:
            continue
        if lst is not None:
            if line.startswith("]"):
                lst = None
            else:
                lst.append(_scalar(line))
            continue
        if line.endswith("{"):
            node = {}
            cur[line[:-1].strip()] = node
            stack.append(cur)
            cur = node
        elif line == "}":
            cur = stack.pop()
        elif line.endswith("{}"):
            cur[line[:-2].split("=")[0].strip()] = {}
        else:
            k, v = [t.strip() for t in line.split("=", 1)]
            if v == "[":
                lst = cur[k] = []
            elif v == "[]":
                cur[k] = []
            else:
                cur[k] = _scalar(v)
    return root


def _scalar(v):
    v = v.strip().strip('"')
    if v in ("true", "false"):
        return v == "true"
    try:
        return int(v)
    except ValueError:
        try:
            return float(v)
        except ValueError:
            return v


def test_conf_presets_equal_the_shipped_run_configs():
    import os
    """utils/conf.py::model_conf / run_conf against EVERY run config the reference ships (code/confs/**/*.conf): the model subtree key by
    key, image size, loss block, loop counts.  Needs the reference checkout (build container only)."""
    import glob
    ref = "/root/reference/code/confs"
    files = sorted(glob.glob(os.path.join(ref, "**", "*.conf"), recursive=True))
    if not files:
        pytest.skip("reference checkout not present")
    from nicer_slam_amd.utils.conf import model_conf, run_conf
    assert len(files) == 23
    for path in files:
        conf = _parse_conf(open(path).read())
        # (the two demo files: runconf_demo_1 is an Azure-family conf -- 720 x 1280, coarse radius 1.0 --, runconf_demo_2 a Replica-family one)
        family = "7scenes" if "7scenes" in path else "azure" if ("azure" in path or path.endswith("demo_1.conf")) else "replica"
        mine = dict(model_conf(family))
        theirs = conf["model"]
        for net in ("coarse", "fine"):
            want = dict(theirs["implicit_network"][net])
            got = dict(mine["implicit_network"][net])
            assert got == want, (path, net, {k: (got.get(k), want.get(k)) for k in set(got) | set(want) if got.get(k) != want.get(k)})
        for key in ("feature_vector_size", "scene_bounding_sphere", "use_warp_loss", "mapping_patchsizes", "tracking_patchsizes",
                    "sampling_method", "density_method", "rendering_network", "ray_sampler"):
            assert mine[key] == theirs[key], (path, key, mine[key], theirs[key])
        rc = run_conf(family)
        assert list(rc["img_res"]) == conf["dataset"]["img_res"], path
        assert {k: float(v) if isinstance(v, (int, float)) and not isinstance(v, bool) else v for k, v in rc["loss"].items()} == \
               {k: float(v) if isinstance(v, (int, float)) and not isinstance(v, bool) else v for k, v in conf["loss"].items()}, (path, rc["loss"], conf["loss"])
        t, m, tr = conf["SLAM"]["tracking"], conf["SLAM"]["mapping"], conf["train"]
        assert rc["const_speed_assumption"] == t.get("const_speed_assumption", False)
        assert (rc["tracking_lr"], rc["BA_cam_lr"], rc["keyframe_every"], rc["mapping_every_frame"], rc["mapping_window_size"]) == \
               (t["lr"], m["BA_cam_lr"], m["keyframe_every"], m["mapping_every_frame"], m["mapping_window_size"]), path
        assert (rc["learning_rate"], rc["lr_factor_for_coarse_grid"], rc["lr_factor_for_fine_grid"], rc["lr_factor_for_color_grid"],
                rc["tracking_num_pixels"]) == (tr["learning_rate"], tr["lr_factor_for_coarse_grid"], tr["lr_factor_for_fine_grid"],
                                               tr["lr_factor_for_color_grid"], tr["tracking_num_pixels"]), path
