"""Fused mapping engine (parameter gradients from the MAP backward kernels, fused/mapping.py) vs the reference goldens
and vs the composed engine.  Tolerances: fp32, sums of ~1e5 atomically accumulated terms -> 2e-4 of the largest entry
of each gradient tensor (same bar as the composed engine's golden test)."""
import numpy as np
import pytest
import torch

from helpers import load, tt, draws_of, golden_objective, assert_close, assert_outputs_close, face_samples
from test_model_cpu import build_model

pytestmark = pytest.mark.gpu

FROZEN = "implicit_network.fine.lin"     # fine SDF MLP: pretrained, not in the reference's optimizer list


def _run(fx, engine, ground_truth=None):
    from nicer_slam_amd.utils.general import camera_from_tensor_torch as get_camera_from_tensor   # the reference's op order: golden comparison (on-face far samples)
    model = build_model(fx).cuda()
    model.freeze_fine_mlp()
    model.engine = engine
    mode, stage, cstage = str(fx["meta_mode"]), str(fx["meta_stage"]), str(fx["meta_color_stage"])
    model.train(True)
    model.voxels = tt(fx["in_voxels"]).cuda()
    model.draws = draws_of(fx, "cuda")
    model.draws["z_vals_override"] = tt(fx["out_z_vals"]).cuda()
    cam = tt(fx["in_cam"]).cuda().requires_grad_(True)
    pose = get_camera_from_tensor(cam)
    out = model({"intrinsics": tt(fx["in_K"]).cuda(), "uv": tt(fx["in_uv"]).cuda(), "pose": pose},
                torch.arange(pose.shape[0], device="cuda"), ground_truth or {}, mode=mode, stage=stage,
                color_stage=cstage, frame_idx=1)
    return model, cam, out


@pytest.mark.parametrize("name", ["full_mapping", "full_mapping_coarse_base", "full_mapping_rw", "full_mapping_rw_coarse",
                                  "full_mapping_7scenes", "full_mapping_7scenes_coarse_base"])
def test_fused_mapping_vs_reference_goldens(name):
    fx = load(name)
    model, cam, out = _run(fx, "fused")
    assert model.last_engine == "fused"
    # The far sample of a ray sits exactly ON the unit-cube face (far bound = cube exit), where the grids' in-range test is decided by
    # the last ulp of o + z d; the fused ray generator and torch differ there by design (both are the reference's formula).  Such
    # samples are excluded from the per-sample colour comparison, and an sdf that shows the other decision is counted (helpers.face_flips).
    on_face = face_samples(fx)
    assert float(on_face.float().mean()) < 0.07
    keys = ("depth_vals", "sdf", "weights", "rgb_values", "depth_values", "entropy", "normal_map", "grad_theta", "grad_theta_nei")
    assert_outputs_close(out, fx, keys)
    assert_close(out["rgb"].detach().cpu()[~on_face], tt(fx["out_rgb"])[~on_face].numpy(), 2e-5, 1e-4, "rgb")
    assert_close(model.voxels, fx["out_voxels"], 0, 0, "voxels")
    golden_objective(out, fx, "mapping").backward()
    assert_close(cam.grad, fx["grad_cam"], 2e-6, 1e-3, "grad_cam")
    checked = 0
    for n, p in model.named_parameters():
        if n.startswith(FROZEN):
            assert p.grad is None
            continue
        ref = fx["grad_" + n]
        if ref.size == 0:
            assert p.grad is None or float(p.grad.abs().max()) == 0, n
        else:
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            assert_close(g, ref, 1e-6 + 2e-4 * float(np.abs(ref).max()), 1e-3, "grad " + n)
            checked += 1
    assert checked >= 10


@pytest.mark.parametrize("name", ["full_tracking", "full_mapping", "full_mapping_coarse_base", "full_tracking_rw",
                                  "full_mapping_rw", "full_mapping_rw_coarse", "full_tracking_7scenes", "full_mapping_7scenes",
                                  "full_mapping_7scenes_coarse_base"])
def test_fused_engine_every_parameter_gradient_like_the_reference(name):
    """The model exactly as volsdf_train.py builds it -- every parameter requires grad, nothing frozen -- with the two
    skip-policies switched off (tracking_param_grads, fine_mlp_grads): the fused engine then produces what the reference's
    autograd produces, tracking and mapping alike, incl. the fine SDF MLP (976 emission rows from k_sdfnet_bwd<fine, MAP>).
    Every named parameter's gradient vs the reference golden."""
    from nicer_slam_amd.utils.general import camera_from_tensor_torch as get_camera_from_tensor   # the reference's op order: golden comparison (on-face far samples)
    fx = load(name)
    model = build_model(fx).cuda()
    assert all(p.requires_grad for p in model.parameters())
    model.engine = "fused"
    model.tracking_param_grads = True
    model.fine_mlp_grads = True
    mode, stage, cstage = str(fx["meta_mode"]), str(fx["meta_stage"]), str(fx["meta_color_stage"])
    model.train(True)
    model.voxels = tt(fx["in_voxels"]).cuda()
    model.draws = draws_of(fx, "cuda")
    model.draws["z_vals_override"] = tt(fx["out_z_vals"]).cuda()
    cam = tt(fx["in_cam"]).cuda().requires_grad_(True)
    pose = get_camera_from_tensor(cam)
    out = model({"intrinsics": tt(fx["in_K"]).cuda(), "uv": tt(fx["in_uv"]).cuda(), "pose": pose},
                torch.arange(pose.shape[0], device="cuda"), {}, mode=mode, stage=stage, color_stage=cstage, frame_idx=1)
    assert model.last_engine == "fused"
    golden_objective(out, fx, mode).backward()
    assert_close(cam.grad, fx["grad_cam"], 2e-6, 1e-3, "grad_cam")
    checked = fine = 0
    for n, p in model.named_parameters():
        ref = fx["grad_" + n]
        if ref.size == 0 or float(np.abs(ref).max()) == 0:
            assert p.grad is None or float(p.grad.abs().max()) == 0, n
            continue
        assert p.grad is not None, n
        assert_close(p.grad, ref, 1e-6 + 2e-4 * float(np.abs(ref).max()), 1e-3, "grad " + n)
        checked += 1
        fine += n.startswith(FROZEN)
    assert checked >= 10 and (fine >= 6 or stage == "coarse"), (checked, fine)


def test_default_policy_keeps_the_unmodified_model_on_the_fused_engine():
    """Defaults: a model built like the reference's (all parameters require grad) renders tracking AND mapping on the fused
    engine with engine='auto'; what the reference computes-and-discards is skipped: no parameter gradients in tracking, none
    for the fine SDF MLP in mapping, every optimizer-list parameter gets its gradient."""
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    fx = load("full_mapping")
    model = build_model(fx).cuda().train(True)
    assert model.engine == "auto" and all(p.requires_grad for p in model.parameters())
    model.voxels = tt(fx["in_voxels"]).cuda()
    model.draws = draws_of(fx, "cuda")
    model.draws["z_vals_override"] = tt(fx["out_z_vals"]).cuda()
    K, uv = tt(fx["in_K"]).cuda(), tt(fx["in_uv"]).cuda()
    cam = tt(fx["in_cam"]).cuda().requires_grad_(True)
    out = model({"intrinsics": K, "uv": uv, "pose": get_camera_from_tensor(cam)}, torch.arange(uv.shape[0], device="cuda"),
                {}, mode="tracking", frame_idx=1)
    assert model.last_engine == "fused"
    out["rgb_values"].abs().mean().backward()
    assert cam.grad is not None and float(cam.grad.abs().max()) > 0
    assert all(p.grad is None for p in model.parameters())
    cam.grad = None
    out = model({"intrinsics": K, "uv": uv, "pose": get_camera_from_tensor(cam)}, torch.arange(uv.shape[0], device="cuda"),
                {}, mode="mapping", stage="fine", color_stage="highfreq", frame_idx=1)
    assert model.last_engine == "fused"
    golden_objective(out, fx, "mapping").backward()
    for n, p in model.named_parameters():
        if n.startswith(FROZEN):
            assert p.grad is None, n
        else:
            ref = fx["grad_" + n]
            if ref.size and float(np.abs(ref).max()) > 0:
                assert_close(p.grad, ref, 1e-6 + 2e-4 * float(np.abs(ref).max()), 1e-3, "grad " + n)


def test_fused_mapping_with_warp_block():
    """Mapping incl. the patch-warp block: the rendered depth feeds torch ops downstream of the fused Function."""
    fx = load("full_mapping_warp")
    gt = {"full_rgb": tt(fx["in_full_rgb"]).cuda(), "full_depth": tt(fx["in_full_depth"]).cuda()}
    grads = {}
    for engine in ("fused", "composed"):
        model, cam, out = _run(fx, engine, gt)
        assert model.last_engine == engine
        loss = (out["rgb_values"].reshape(-1, 3) - tt(fx["gt_rgb"]).cuda()).abs().mean()
        for ps, (gt_w, samp, mask, ray_mask) in out["warp_output"].items():
            loss = loss + 0.5 * ((gt_w - samp).abs().sum(-1) * mask.float()).sum() / (mask.float().sum() + 1)
        loss = loss + 0.1 * ((out["grad_theta"].norm(2, dim=1) - 1) ** 2).mean()
        loss.backward()
        grads[engine] = (float(loss), cam.grad.clone(), {n: p.grad for n, p in model.named_parameters()})
    assert abs(grads["fused"][0] - float(fx["out_loss"]) - (grads["composed"][0] - float(fx["out_loss"]))) < 1e-5
    ref_cam = grads["composed"][1]
    assert_close(grads["fused"][1], ref_cam.cpu().numpy(), 2e-3 * float(ref_cam.abs().max()), 2e-3, "grad_cam")
    for n, g in grads["composed"][2].items():
        if n.startswith(FROZEN) or g is None:
            continue
        f = grads["fused"][2][n]
        assert f is not None, n
        assert_close(f, g.cpu().numpy(), 1e-6 + 3e-4 * float(g.abs().max()), 1e-3, "grad " + n)


def test_fused_mapping_larger_batch_vs_composed():
    """512 rays x 98 samples + 11k eikonal points, random weights/tables: every trainable gradient, both engines."""
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    torch.manual_seed(3)
    model = SLAMNetwork(replica_model_conf(use_warp_loss=False)).cuda().freeze_fine_mlp()
    with torch.no_grad():
        for enc in (model.implicit_network.coarse.encoding, model.implicit_network.fine.encoding,
                    model.rendering_network.encoding):
            enc.embeddings.uniform_(-0.05, 0.05)
    model.train(True)
    R = 512
    uv = torch.rand(1, R, 2, device="cuda") * torch.tensor([1200.0, 680.0], device="cuda")
    K = torch.eye(4, device="cuda")[None].clone()
    K[0, 0, 0] = K[0, 1, 1] = 600.0
    K[0, 0, 2], K[0, 1, 2] = 599.5, 339.5
    # the two engines are compared from ONE sample set: z_vals of the first run are handed to the second, and the index of the
    # near-surface eikonal sample is pinned so that it is the same element of that set (two correct samplers differ where the
    # inverse CDF is ill-conditioned; the samplers themselves are compared in CDF space in test_sampler_gpu.py)
    samp = model.ray_sampler
    draws = {"eik_idx": torch.randint(samp.N_samples + 2 + samp.N_samples_extra, (R,), device="cuda")}
    vox0 = torch.randint(0, 50, (64, 64, 64), device="cuda").float()
    res = {}
    for engine in ("composed", "fused"):
        model.engine = engine
        model.zero_grad(set_to_none=True)
        model.voxels = vox0.clone()
        model.draws = dict(draws)
        torch.manual_seed(11)                                   # same device-generator draws for both engines
        cam = torch.tensor([[1.0, 0.02, -0.03, 0.01, 0.1, -0.05, 0.2]], device="cuda", requires_grad=True)
        out = model({"intrinsics": K, "uv": uv, "pose": get_camera_from_tensor(cam)}, torch.arange(1, device="cuda"), {},
                    mode="mapping", stage="fine", color_stage="highfreq", frame_idx=1)
        assert model.last_engine == engine
        loss = (out["rgb_values"] - 0.4).abs().mean() + 0.3 * (out["depth_values"] - 1.5).abs().mean() \
            + 0.2 * (out["normal_map"] - 0.1).abs().mean() + 0.1 * ((out["grad_theta"].norm(2, dim=1) - 1) ** 2).mean() \
            + 0.05 * (out["grad_theta"] - out["grad_theta_nei"]).norm(2, dim=-1).mean() + 0.01 * out["entropy"]
        loss.backward()
        res[engine] = (float(loss), cam.grad.clone(), {n: (None if p.grad is None else p.grad.clone())
                                                      for n, p in model.named_parameters()})
        if "z_vals_override" not in draws:
            draws["z_vals_override"] = out["z_vals"].detach()
    assert abs(res["fused"][0] - res["composed"][0]) < 1e-5 * abs(res["composed"][0]) + 1e-6
    c_cam = res["composed"][1]
    assert_close(res["fused"][1], c_cam.cpu().numpy(), 1e-3 * float(c_cam.abs().max()), 1e-3, "grad_cam")
    n_checked = 0
    for n, g in res["composed"][2].items():
        if n.startswith(FROZEN) or g is None:
            continue
        f = res["fused"][2][n]
        assert f is not None, n
        assert_close(f, g.cpu().numpy(), 1e-7 + 3e-4 * float(g.abs().max()), 1e-3, "grad " + n)
        n_checked += 1
    assert n_checked >= 12


def test_update_voxels_kernel_vs_torch_index_add():
    """nsa_update_voxels vs SLAMNetwork.update_voxels (torch index_add_) on 200k samples incl. out-of-range ones: exact."""
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.fused import mapping
    torch.manual_seed(5)
    model = SLAMNetwork(replica_model_conf(use_warp_loss=False)).cuda()
    R, S = 2048, 98
    o = (torch.rand(R, 3, device="cuda") - 0.5) * 0.6
    d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=-1)
    z = torch.sort(torch.rand(R, S, device="cuda") * 1.8, dim=1).values
    model.voxels = torch.randint(0, 7, (64, 64, 64), device="cuda").float()
    ref = SLAMNetwork(replica_model_conf(use_warp_loss=False)).cuda()
    ref.voxels = model.voxels.clone()
    x = (o.unsqueeze(1) + z.unsqueeze(2) * d.unsqueeze(1)).reshape(-1, 3)
    assert 0.02 < float((x.abs() > 0.99).any(1).float().mean()) < 0.9
    ref.update_voxels(x)
    mapping.update_voxels(model, o, d, z)
    assert torch.equal(model.voxels, ref.voxels)
    assert float(model.voxels.sum()) > float(7 * 64 ** 3 / 2)


@pytest.mark.parametrize("n", [1, 5, 1027, (1 << 20) + 3])
def test_fused_adam_vs_torch_adam(n):
    """nicer_slam_amd.optim.Adam (one HIP pass) vs torch.optim.Adam(betas=(0.9,0.99), eps=1e-15): 6 steps, few-ulp."""
    from nicer_slam_amd.optim import Adam
    torch.manual_seed(n)
    p0 = torch.randn(n, device="cuda") * 0.1
    a = torch.nn.Parameter(p0.clone())
    b = torch.nn.Parameter(p0.clone())
    oa = Adam([{"params": [a], "lr": 0.04}], betas=(0.9, 0.99), eps=1e-15)
    ob = torch.optim.Adam([{"params": [b], "lr": 0.04}], betas=(0.9, 0.99), eps=1e-15)
    for it in range(6):
        g = torch.randn(n, device="cuda") * (10.0 ** (it - 3))
        g[::3] = 0.0                                          # untouched table rows have exactly-zero gradients
        a.grad, b.grad = g.clone(), g.clone()
        v0 = a._version
        oa.step()
        ob.step()
        assert a._version > v0
        assert_close(a.detach(), b.detach().cpu().numpy(), 1e-7, 2e-6, f"param after step {it + 1}")
    sa, sb = oa.state[a], ob.state[b]
    assert float(sa["step"]) == float(sb["step"]) == 6
    # entries where the moments cancel carry the rounding of the larger terms: tolerance relative to the tensor's scale
    assert_close(sa["exp_avg"], sb["exp_avg"].cpu().numpy(), 3e-7 * float(sb["exp_avg"].abs().max()), 2e-6, "exp_avg")
    assert_close(sa["exp_avg_sq"], sb["exp_avg_sq"].cpu().numpy(), 3e-7 * float(sb["exp_avg_sq"].abs().max()), 2e-6,
                 "exp_avg_sq")


@pytest.mark.parametrize("n", [5, 1027, (1 << 20) + 3, (1 << 26) + 8])
def test_adam_none_grad_zeros_is_the_reference_environments_zero_grad(n):
    """optim.Adam(none_grad="zeros"): a parameter whose .grad is None (what optimizer.zero_grad() leaves under torch >= 2.0) but which was
    stepped before takes the step torch 1.11 -- the reference's environment, env_yamls/nicer-slam.yaml:62 -- takes on the ZERO tensor its
    zero_grad() leaves (volsdf_train.py:547): moments decay, the parameter moves along its momentum.  Expected values: torch.optim.Adam fed
    explicit zero gradients.  Sizes: multi-tensor launch (5, 1027), table kernel (2^20 + 3), non-temporal table kernel (2^26 + 8).
    "skip" (default) must leave such a parameter and its state untouched, and a never-stepped parameter is skipped in both modes."""
    from nicer_slam_amd.optim import Adam
    torch.manual_seed(n % 1000)
    p0 = torch.randn(n, device="cuda") * 0.1
    a, b, c = (torch.nn.Parameter(p0.clone()) for _ in range(3))
    fresh = torch.nn.Parameter(p0[:7].clone())
    oa = Adam([{"params": [a, fresh], "lr": 0.04}], betas=(0.9, 0.99), eps=1e-15, none_grad="zeros")
    ob = torch.optim.Adam([{"params": [b], "lr": 0.04}], betas=(0.9, 0.99), eps=1e-15)
    oc = Adam([{"params": [c], "lr": 0.04}], betas=(0.9, 0.99), eps=1e-15)
    assert oc.none_grad == "skip"
    with pytest.raises(ValueError):
        Adam([a], none_grad="zero")
    oa.step()                                                 # nothing has a gradient or a state yet: nothing happens
    assert torch.equal(a.detach(), p0) and not oa.state.get(a) and not oa.state.get(fresh)
    schedule = [True, True, False, False, False, True, False]     # gradient this step?
    for it, has in enumerate(schedule):
        g = torch.randn(n, device="cuda") * (10.0 ** (it % 3 - 2))
        g[::3] = 0.0
        a.grad, c.grad = (g.clone(), g.clone()) if has else (None, None)
        b.grad = g.clone() if has else torch.zeros_like(b)
        c_before = c.detach().clone()
        v0 = a._version
        oa.step()
        ob.step()
        oc.step()
        assert a._version > v0
        assert_close(a.detach(), b.detach().cpu().numpy(), 1e-7, 2e-6, f"param after step {it + 1}")
        if not has:
            assert torch.equal(c.detach(), c_before)          # "skip": untouched
            assert float((a.detach() - c_before).abs().max()) > 0 or it == 0
    sa, sb, sc = oa.state[a], ob.state[b], oc.state[c]
    assert float(sa["step"]) == float(sb["step"]) == len(schedule) and float(sc["step"]) == sum(schedule)
    assert not oa.state.get(fresh)
    assert_close(sa["exp_avg"], sb["exp_avg"].cpu().numpy(), 3e-7 * float(sb["exp_avg"].abs().max()), 2e-6, "exp_avg")
    assert_close(sa["exp_avg_sq"], sb["exp_avg_sq"].cpu().numpy(), 3e-7 * float(sb["exp_avg_sq"].abs().max()), 2e-6, "exp_avg_sq")
    # the options survive pickling and a state_dict round trip (device `step` tensors of a capturable torch state are moved to the host)
    import pickle
    o2 = pickle.loads(pickle.dumps(oa))
    assert o2.none_grad == "zeros" and o2.consume_table_grads == oa.consume_table_grads
    del o2.__dict__["none_grad"], o2.__dict__["consume_table_grads"]
    o2.__setstate__(torch.optim.Optimizer.__getstate__(o2))       # a pickle made before the options existed
    assert o2.none_grad == "skip" and o2.consume_table_grads is False
    sd = ob.state_dict()
    for st in sd["state"].values():
        st["step"] = st["step"].to("cuda")
    o3 = Adam([{"params": [b], "lr": 0.04}], betas=(0.9, 0.99), eps=1e-15)
    o3.load_state_dict(sd)
    assert o3.state[b]["step"].device.type == "cpu" and float(o3.state[b]["step"]) == len(schedule)


def test_sharded_adam_single_rank_uses_hip_stepper():
    """ShardedAdam without a process group (world 1) must step exactly like nicer_slam_amd.optim.Adam (same kernel)."""
    from nicer_slam_amd.optim import Adam
    from nicer_slam_amd.dist import ShardedAdam
    torch.manual_seed(2)
    p0 = [torch.randn(70001, 2, device="cuda"), torch.randn(33, device="cuda")]
    a = [torch.nn.Parameter(t.clone()) for t in p0]
    b = [torch.nn.Parameter(t.clone()) for t in p0]
    oa = ShardedAdam([{"params": a, "lr": 0.01}], betas=(0.9, 0.99), eps=1e-15)
    ob = Adam([{"params": b, "lr": 0.01}], betas=(0.9, 0.99), eps=1e-15)
    for it in range(3):
        for x, y in zip(a, b):
            g = torch.randn_like(x)
            x.grad, y.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for x, y in zip(a, b):
        assert torch.equal(x.detach(), y.detach())
    with pytest.raises(RuntimeError):
        c = torch.nn.Parameter(torch.randn(4))
        c.grad = torch.randn(4)
        ShardedAdam([c]).step()               # CPU tensors: no fallback


def test_mapping_step_with_full_slam_loss_fused_vs_composed():
    """One mapping iteration as the training loop assembles it -- FrameFeed batch -> SLAMNetwork(mode='mapping') ->
    SLAMLoss with the shipped Replica weights (rgb, SSI depth, normal L1/cos, eikonal, smooth) -> backward -> Adam --
    on the fused engine with the HIP optimizer vs the composed engine with torch.optim.Adam: same loss terms, same
    trainable gradients, same parameters after the step."""
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.model.loss import SLAMLoss
    from nicer_slam_amd.feed import FrameFeed
    from nicer_slam_amd.optim import Adam
    H, W, n_pix, frames = 30, 40, 96, 3
    torch.manual_seed(21)
    feed = FrameFeed((H, W), device="cuda", scene_scale=1.0)
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = 35.0
    K[0, 2], K[1, 2] = W / 2 - 0.5, H / 2 - 0.5
    for f in range(frames):
        pose = torch.eye(4)
        pose[:3, 3] = torch.tensor([0.05 * f, 0.02, -0.2 + 0.03 * f])
        feed.add_frame(f, rgb=torch.rand(H * W, 3), depth=torch.rand(H * W, 1) * 0.02 + 0.01,
                       normal=torch.nn.functional.normalize(torch.randn(H * W, 3), dim=-1),
                       gt_depth=torch.rand(H * W, 1) * 2 + 0.5, intrinsics=K, pose=pose)
    feed.change_sampling_idx(n_pix, generator=torch.Generator(device="cuda").manual_seed(3))
    indices, model_input, gt = feed.batch(range(frames))

    class DS:
        data_dir = "synthetic"
    results = {}
    for engine in ("composed", "fused"):
        torch.manual_seed(5)
        model = SLAMNetwork(replica_model_conf(use_warp_loss=False)).cuda().freeze_fine_mlp()
        with torch.no_grad():
            for enc in (model.implicit_network.coarse.encoding, model.implicit_network.fine.encoding,
                        model.rendering_network.encoding):
                enc.embeddings.uniform_(-0.05, 0.05)
        model.train(True)
        model.engine = engine
        groups = [{"params": list(model.implicit_network.fine.grid_parameters()), "lr": 0.04},
                  {"params": list(model.implicit_network.coarse.grid_parameters()), "lr": 0.04},
                  {"params": list(model.rendering_network.grid_parameters()), "lr": 0.01},
                  {"params": list(model.rendering_network.mlp_parameters()) + list(model.implicit_network.coarse.mlp_parameters()),
                   "lr": 0.002}]
        opt = (Adam if engine == "fused" else torch.optim.Adam)(groups, betas=(0.9, 0.99), eps=1e-15)
        crit = SLAMLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, train_dataset=DS(), scan_id=1,
                        assign_scale_shift_init=True, smooth_weight=0.005, depth_weight=0.1, normal_l1_weight=0.05,
                        normal_cos_weight=0.05)
        torch.manual_seed(9)                                    # same device-generator draws for both engines
        if "z" in results:
            model.draws = {"z_vals_override": results["z"]}
        out = model(model_input, indices.cuda(), gt, mode="mapping", stage="fine", color_stage="highfreq", frame_idx=5)
        assert model.last_engine == engine
        results.setdefault("z", out["z_vals"].detach())
        terms = crit(out, gt, keyframe_list=None, frame_idx=5, stage="fine")
        opt.zero_grad()
        terms["loss"].backward()
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        opt.step()
        results[engine] = ({k: float(v) for k, v in terms.items()}, grads,
                           {n: p.detach().clone() for n, p in model.named_parameters()})
    tc, gc, pc = results["composed"]
    tf_, gf, pf = results["fused"]
    for k in tc:
        assert abs(tc[k] - tf_[k]) <= 2e-5 * max(1.0, abs(tc[k])), (k, tc[k], tf_[k])
    assert tc["depth_loss"] > 0 and tc["normal_l1"] > 0 and tc["eikonal_loss"] > 0 and tc["smooth_loss"] > 0
    checked = 0
    for n, g in gc.items():
        if n.startswith(FROZEN):
            continue
        assert n in gf, n
        # 1e-6 floor: bias gradients are sums of ~3e5 signed fp32 terms that largely cancel (summation-order noise)
        assert_close(gf[n], g.cpu().numpy(), 1e-6 + 4e-4 * float(g.abs().max()), 1e-3, "grad " + n)
        checked += 1
    assert checked >= 12
    for n in pc:                                               # Adam's first step is +-lr wherever the gradient is non-zero:
        if n.startswith(FROZEN):                                # compare where both engines agree on a clearly non-zero gradient
            continue
        if n in gc:
            big = gc[n].abs() > 1e-3 * gc[n].abs().max()
            assert_close(pf[n][big], pc[n][big].cpu().numpy(), 1e-6, 1e-5, "param after step " + n)


def test_flow_block_downstream_of_fused_function():
    """Optical-flow reprojection (network.py:153-165: rendered depth -> 3-D point -> projection into frame j) computed by
    torch ops on the fused Function's depth output: flow and the gradients it sends back, fused vs composed engine."""
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    torch.manual_seed(13)
    bs, n = 3, 64
    K = torch.eye(4, device="cuda")
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    K = K[None].repeat(bs, 1, 1)
    idx = torch.randint(680 * 1200, (n,), device="cuda")
    uv = torch.stack([(idx % 1200).float(), (idx // 1200).float()], -1)[None].repeat(bs, 1, 1)
    edges = (torch.tensor([0, 1], device="cuda"), torch.tensor([1, 2], device="cuda"), None, None)
    target = torch.randn(2, n, 2, device="cuda") * 3
    res = {}
    for engine in ("composed", "fused"):
        torch.manual_seed(5)
        model = SLAMNetwork(replica_model_conf(use_warp_loss=False)).cuda().freeze_fine_mlp()
        with torch.no_grad():
            for enc in (model.implicit_network.coarse.encoding, model.implicit_network.fine.encoding,
                        model.rendering_network.encoding):
                enc.embeddings.uniform_(-0.05, 0.05)
        model.train(True)
        model.engine = engine
        if "z" in res:
            model.draws = {"z_vals_override": res["z"]}
        torch.manual_seed(9)
        cams = torch.tensor([[1.0, 0.01, -0.02, 0.0, 0.1, 0.0, -0.2], [1.0, 0.02, 0.0, 0.01, 0.12, 0.01, -0.2],
                             [1.0, -0.01, 0.02, 0.0, 0.08, -0.01, -0.22]], device="cuda", requires_grad=True)
        out = model({"intrinsics": K, "uv": uv, "pose": get_camera_from_tensor(cams)}, torch.arange(bs, device="cuda"),
                    {"edges": edges}, mode="mapping", stage="fine", color_stage="highfreq", frame_idx=5)
        assert model.last_engine == engine and out["flow"].shape == (2, n, 2)
        res.setdefault("z", out["z_vals"].detach())
        loss = (out["flow"] - target).abs().mean() + (out["rgb_values"] - 0.5).abs().mean()
        loss.backward()
        res[engine] = (out["flow"].detach(), cams.grad.clone(),
                       {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    assert_close(res["fused"][0], res["composed"][0].cpu().numpy(), 2e-3, 1e-4, "flow (pixels)")
    gc = res["composed"][1]
    assert_close(res["fused"][1], gc.cpu().numpy(), 2e-3 * float(gc.abs().max()), 2e-3, "grad_cam")
    for k, g in res["composed"][2].items():
        if k.startswith(FROZEN):
            continue
        assert_close(res["fused"][2][k], g.cpu().numpy(), 1e-6 + 4e-4 * float(g.abs().max()), 1e-3, "grad " + k)


@pytest.mark.parametrize("case", [dict(a=(5, 140), M=64, b=(70, 200), N=96), dict(a=(3,), M=64, b=(80,), N=130),
                                  dict(a=(9,), M=3, b=(20,), N=64), dict(a=(11,), M=64, b=None, N=0),
                                  dict(a=(0, 64), M=33, b=(128, 200), N=31)])
def test_emit_gemm_vs_float64(case):
    """nsa_emit_gemm (weight + bias gradients from emission rows) against a float64 product of the same rows: fp32-faithful
    (error of an fp32 dot product, not of bf16 operands), deterministic, ragged M / N, one and two product pairs."""
    from nicer_slam_amd.fused.mapping import emit_gemm, emit_ld
    g = torch.Generator().manual_seed(3)
    P = 3 * 4096 - 1234
    ld = emit_ld(P)
    emit = torch.zeros(300, ld)
    emit[:, :P] = torch.randn(300, P, generator=g) * torch.rand(300, 1, generator=g) * 3
    dev = emit.cuda()
    a, b, M, N = case["a"], case["b"], case["M"], case["N"]
    out = emit_gemm(dev, a, M, b, N)
    again = emit_gemm(dev, a, M, b, N)
    assert torch.equal(out, again)
    e64 = emit.double()
    ref = torch.zeros(M, N + 1, dtype=torch.float64)
    scale = torch.zeros(M, N + 1, dtype=torch.float64)
    for j, a0 in enumerate(a):
        if N:
            ref[:, :N] += e64[a0:a0 + M] @ e64[b[j]:b[j] + N].T
            scale[:, :N] += e64[a0:a0 + M].abs() @ e64[b[j]:b[j] + N].abs().T
    ref[:, N] = e64[a[0]:a[0] + M].sum(1)
    scale[:, N] = e64[a[0]:a[0] + M].abs().sum(1)
    err = (out.double().cpu() - ref).abs() / scale.clamp_min(1e-30)
    print("emit_gemm max error / sum|ab|:", float(err.max()))
    assert float(err.max()) < 1e-6, float(err.max())      # fp32 accumulation over ~11k terms (x2 pairs)


def test_flat_weight_norm_kernels_vs_torch_weight_norm():
    """fused/pack.py::FlatWeightNorm (nsa_weight_norm_flat / _backward: one launch per direction) vs per-layer torch._weight_norm
    + reshape + cat and its autograd, on the three MLP shapes of the model (base_networks.py:137-141, 376-379)."""
    from nicer_slam_amd.fused import pack
    fx = load("full_mapping")
    model = build_model(fx).cuda()
    nets = [model.implicit_network.coarse, model.implicit_network.fine, model.rendering_network]
    torch.manual_seed(3)
    for net in nets:
        with torch.no_grad():
            for p in net.mlp_parameters():
                p.add_(0.1 * torch.randn_like(p))
        wn = pack._wn_params(net)
        assert wn is not None and len(wn) == 3 * (net.num_layers - 1)
        flat = pack.flat_params(net)
        parts = []
        for l in range(net.num_layers - 1):
            lin = getattr(net, "lin" + str(l))
            parts += [torch._weight_norm(lin.weight_v, lin.weight_g, 0).reshape(-1), lin.bias.reshape(-1)]
        ref = torch.cat(parts + [parts[0].new_zeros(1)])
        assert flat.shape == ref.shape and float(flat[-1]) == 0.0
        assert_close(flat.detach(), ref.detach().cpu().numpy(), 1e-7, 2e-6, "flat effective parameters")
        c = torch.randn_like(ref)
        params = list(net.mlp_parameters())
        g_kernel = torch.autograd.grad(flat, params, c)
        g_torch = torch.autograd.grad(ref, params, c)
        for p, a, b in zip(params, g_kernel, g_torch):
            assert a.shape == p.shape
            assert_close(a, b.cpu().numpy(), 2e-6 * float(b.abs().max()) + 1e-9, 2e-5, "weight-norm backward")
        # detached: computed once per parameter version, recomputed after an in-place update
        with torch.no_grad():
            d0 = pack.flat_params(net)
            assert pack.flat_params(net) is d0
            params[0].mul_(1.5)
            d1 = pack.flat_params(net)
        assert d1 is not d0 and not torch.equal(d0, d1)


@pytest.mark.parametrize("consume", [False, True])
def test_adam_consumes_the_persistent_table_gradient_buffer(consume):
    """fused/tablegrad.py: the table's .grad is the engine's persistent buffer; the HIP Adam steps like torch.optim.Adam on it and
    either clears it itself (nsa_adam_table_step_clear) or leaves that to the engine (default: nsa_fill_zero when the next backward
    acquires it); either way the next backward finds it clean.  A gradient that is
    NOT that buffer is left untouched (torch semantics)."""
    from nicer_slam_amd.optim import Adam
    from nicer_slam_amd.fused import tablegrad
    torch.manual_seed(5)
    n = (1 << 18) + 6
    p0 = torch.randn(n, 2, device="cuda") * 0.1
    a, b = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p0.clone())
    oa = Adam([{"params": [a], "lr": 0.04}], betas=(0.9, 0.99), eps=1e-15, consume_table_grads=consume)
    ob = torch.optim.Adam([{"params": [b], "lr": 0.04}], betas=(0.9, 0.99), eps=1e-15)
    ptr = None
    for it in range(4):
        oa.zero_grad()
        assert a.grad is None
        buf = tablegrad.target(a)                       # what a MAP backward kernel is handed
        assert a.grad is buf and buf.shape == a.shape
        assert float(buf.abs().max()) == 0.0, "the buffer must come back clean without a fill"
        assert ptr is None or buf.data_ptr() == ptr
        ptr = buf.data_ptr()
        g = torch.randn(n, 2, device="cuda") * (10.0 ** (it - 2))
        g[::3] = 0.0
        buf.add_(g)
        assert tablegrad.target(a) is buf               # second pass of the same iteration accumulates in place
        b.grad = g.clone()
        oa.step()
        ob.step()
        assert_close(a.detach(), b.detach().cpu().numpy(), 1e-7, 2e-6, f"param after step {it + 1}")
        assert (float(buf.abs().max()) == 0.0) == consume      # cleared by the step kernel / left for the engine to clear
    # a caller-owned gradient is stepped on but not cleared
    oa.zero_grad()
    a.grad = torch.ones_like(a)
    oa.step()
    assert float(a.grad.min()) == 1.0
    # left dirty (no step): the next acquisition clears it
    oa.zero_grad()
    tablegrad.target(a).add_(1.0)
    oa.zero_grad()
    assert float(tablegrad.target(a).abs().max()) == 0.0


def test_table_gradients_in_place_vs_through_autograd():
    """The default in-place table gradients (MAP kernels scatter into param.grad = a persistent buffer) against the
    NSA_TABLE_GRADS=autograd form (fresh zero-filled gradients returned through autograd) on the full mapping golden: same
    gradients up to the atomics' summation order; two backward passes without zero_grad accumulate; autograd.grad raises."""
    from nicer_slam_amd.fused import tablegrad
    fx = load("full_mapping")
    tables = ("implicit_network.coarse.encoding.embeddings", "implicit_network.fine.encoding.embeddings",
              "rendering_network.encoding.embeddings")
    grads = {}
    assert tablegrad.IN_PLACE
    try:
        for mode in (True, False):
            tablegrad.IN_PLACE = mode
            model, cam, out = _run(fx, "fused")
            golden_objective(out, fx, "mapping").backward()
            named = dict(model.named_parameters())
            grads[mode] = {k: named[k].grad.clone() for k in tables}
            for k in tables:
                assert tablegrad.consumable(named[k], named[k].grad) == mode, k
            if mode:          # a second forward + backward without zero_grad: gradients accumulate like autograd's
                model.voxels = tt(fx["in_voxels"]).cuda()
                model.draws = draws_of(fx, "cuda")
                model.draws["z_vals_override"] = tt(fx["out_z_vals"]).cuda()
                from nicer_slam_amd.utils.general import camera_from_tensor_torch
                out2 = model({"intrinsics": tt(fx["in_K"]).cuda(), "uv": tt(fx["in_uv"]).cuda(),
                              "pose": camera_from_tensor_torch(cam.detach())},
                             torch.arange(out["rgb_values"].shape[0], device="cuda"), {}, mode="mapping",
                             stage=str(fx["meta_stage"]), color_stage=str(fx["meta_color_stage"]), frame_idx=1)
                golden_objective(out2, fx, "mapping").backward()
                for k in tables:
                    ref = 2.0 * grads[True][k]
                    assert_close(named[k].grad, ref.cpu().numpy(), 1e-6 + 2e-4 * float(ref.abs().max()), 1e-3, "accumulated " + k)
                out3 = model({"intrinsics": tt(fx["in_K"]).cuda(), "uv": tt(fx["in_uv"]).cuda(),
                              "pose": camera_from_tensor_torch(cam.detach())},
                             torch.arange(out["rgb_values"].shape[0], device="cuda"), {}, mode="mapping",
                             stage=str(fx["meta_stage"]), color_stage=str(fx["meta_color_stage"]), frame_idx=1)
                with pytest.raises(RuntimeError):
                    torch.autograd.grad(golden_objective(out3, fx, "mapping"), [named[tables[2]]])
    finally:
        tablegrad.IN_PLACE = True
    assert float(grads[False][tables[2]].abs().max()) > 0
    for k in tables:
        ref = grads[False][k]
        assert_close(grads[True][k], ref.cpu().numpy(), 1e-7 + 2e-5 * float(ref.abs().max()), 1e-3, k)


@pytest.mark.parametrize("P", [1, 63, 4097, 100003, 8192 * 98])
@pytest.mark.parametrize("bits", [30, 24, 9])
def test_morton_order_radix_sort_vs_torch_stable_sort(P, bits):
    """nsa_morton_order (in-library LSD radix sort of the Morton keys) = the stable argsort of the keys' top `bits` bits."""
    import ctypes
    from nicer_slam_amd._native import lib, check, PointsDesc
    g = torch.Generator(device="cuda").manual_seed(P + bits)
    pts = (torch.rand(P, 3, device="cuda", generator=g) * 2.2 - 1.1)
    if P > 100:
        pts[::5] = pts[7]                                     # many equal keys: stability matters
    desc = PointsDesc(None, None, None, pts.data_ptr(), P, 0, None)
    st = torch.cuda.current_stream().cuda_stream
    keys = torch.empty(P, device="cuda", dtype=torch.int32)
    check(lib.nsa_morton_keys(ctypes.byref(desc), keys.data_ptr(), st))
    order = torch.full((P,), -1, device="cuda", dtype=torch.int32)
    ws = torch.empty(int(lib.nsa_morton_order_workspace(P)), device="cuda", dtype=torch.int32)
    check(lib.nsa_morton_order(ctypes.byref(desc), order.data_ptr(), ws.data_ptr(), bits, st))
    ref = torch.sort(keys.long() >> (30 - bits), stable=True).indices
    assert torch.equal(order.long(), ref)
    assert lib.nsa_morton_order(ctypes.byref(desc), order.data_ptr(), ws.data_ptr(), 31, st) != 0
