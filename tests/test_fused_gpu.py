"""Fused composite kernels (SDF networks, colour network, compositing; forward and hand-derived backward) vs the
goldens captured from the reference and vs the CPU oracle.  Needs an MI355X.

Tolerances: forward tensors abs 2e-5 / rel 1e-4 (fp32); gradients rel 1e-3 of the tensor's max."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import load, tt, params_of, oracle_config, draws_of, golden_objective, assert_close, face_flips, assert_outputs_close
from test_model_cpu import build_model

pytestmark = pytest.mark.gpu


def _setup(name):
    from nicer_slam_amd.utils.general import camera_from_tensor_torch as get_camera_from_tensor   # the reference's op order: golden comparison (on-face far samples)
    from nicer_slam_amd.utils import rend_util
    fx = load(name)
    model = build_model(fx).cuda()
    model.train(bool(fx["meta_training"]))
    model.voxels = tt(fx["in_voxels"]).cuda()
    for p in model.parameters():
        p.requires_grad_(False)
    cam = tt(fx["in_cam"]).cuda().requires_grad_(True)
    pose = get_camera_from_tensor(cam)
    d, o = rend_util.get_camera_params(tt(fx["in_uv"]).cuda(), pose, tt(fx["in_K"]).cuda())
    bs, n, _ = d.shape
    rays_d = d.reshape(-1, 3)
    rays_o = o.unsqueeze(1).repeat(1, n, 1).reshape(-1, 3)
    return fx, model, cam, pose, rays_o, rays_d


@pytest.mark.parametrize("name", ["full_tracking", "full_tracking_poisson", "full_vis_eval", "full_tracking_rw",
                                  "full_tracking_7scenes"])
def test_stage_by_stage_forward(name):
    """sdf / grad sdf / feature / rgb / weights of the fused kernels at the reference's own sample positions."""
    from oracle import render_ref as R
    from nicer_slam_amd.fused import render as fr
    fx, model, cam, pose, rays_o, rays_d = _setup(name)
    z = tt(fx["out_z_vals"]).cuda()
    with torch.no_grad():
        rgbv, depth, nmap, w, ent, sdf, rgb, grad = fr.composite(model, rays_o.detach(), rays_d.detach(), z, "fine",
                                                               "highfreq")
    # on-face far samples that took the other in-range decision than the golden (helpers.face_flips: counted, bounded; everything
    # else must agree): such a sample is left out of the per-sample checks, its ray out of the per-ray ones
    flips = face_flips(sdf, fx)
    keep_s, keep_r = ~flips, ~flips.any(-1)
    print(f"{name}: {int(flips.sum())} on-face flips")
    assert_close(sdf.cpu()[keep_s], tt(fx["out_sdf"])[keep_s], 2e-5, 1e-4, "sdf")
    assert_close(rgb.cpu()[keep_s], tt(fx["out_rgb"])[keep_s], 2e-5, 1e-4, "rgb per sample")
    assert_close(w.cpu()[keep_r], tt(fx["out_weights"])[keep_r], 2e-5, 1e-4, "weights")
    assert_close(rgbv.cpu()[keep_r], tt(fx["out_rgb_values"]).reshape(-1, 3)[keep_r], 2e-5, 1e-4, "rgb_values")
    # grad sdf and the normal map vs the oracle (the goldens hold only the rotated normal map)
    cfg, params = oracle_config(fx), params_of(fx)
    pts = (rays_o.detach().cpu().unsqueeze(1) + z.cpu().unsqueeze(2) * rays_d.detach().cpu().unsqueeze(1)).reshape(-1, 3)
    s_o, f_o, g_o = R.sdf_outputs(params, cfg, pts.clone(), "fine")
    ks = keep_s.reshape(-1)
    # (the oracle evaluates the torch-expression points: an on-face sample may sit on the other side there as well)
    ok = ks & ((s_o.reshape(-1) - sdf.cpu().reshape(-1)).abs() < 1e-4)
    assert int((ks & ~ok).sum()) <= max(1, int(0.01 * ks.numel()))
    assert_close(grad.cpu()[ok], g_o[ok], 2e-5 * float(g_o.abs().max()), 1e-4, "grad sdf")
    nm = torch.einsum("bij,bni->bnj", pose[:, :3, :3].detach(), nmap.reshape(pose.shape[0], -1, 3))
    assert_close(nm.cpu().reshape(-1, 3)[keep_r], tt(fx["out_normal_map"]).reshape(-1, 3)[keep_r], 2e-5, 1e-4, "normal_map")
    if not bool(flips.any()):
        assert_close(ent.mean(), fx["out_entropy"], 2e-5, 1e-4, "entropy")


def test_feature_vector_hl_layout():
    from oracle import render_ref as R
    from nicer_slam_amd.fused import render as fr, sampler as fs
    from nicer_slam_amd._native import lib, check
    import ctypes
    fx, model, cam, pose, rays_o, rays_d = _setup("full_tracking")
    z = tt(fx["out_z_vals"]).cuda()
    P = z.numel()
    ro, rd = rays_o.detach().contiguous(), rays_d.detach().contiguous()
    pts = fr._pts(ro, rd, z)
    sdf, grad = torch.empty(P, device="cuda"), torch.empty(P, 3, device="cuda")
    feat = torch.empty(fr.hl_size(P), device="cuda")
    imp = model.implicit_network
    for which, acc, nh in (("coarse", 0, 1), ("fine", 1, 3)):
        net = getattr(imp, which)
        g, keep = fs.sdf_grid_desc(model, which)
        check(lib.nsa_sdfnet_forward(ctypes.byref(pts), ctypes.byref(g), fs.packed_sdf(model, which).data_ptr(), acc,
                                     sdf.data_ptr(), grad.data_ptr(), feat.data_ptr(), torch.cuda.current_stream().cuda_stream))
    dense = feat[fr.hl_index(P, "cuda")]
    cfg, params = oracle_config(fx), params_of(fx)
    x = (ro.cpu().unsqueeze(1) + z.cpu().unsqueeze(2) * rd.cpu().unsqueeze(1)).reshape(-1, 3)
    s_o, f_o, g_o = R.sdf_outputs(params, cfg, x.clone(), "fine")
    assert_close(dense, f_o, 2e-5 * float(f_o.abs().max()), 1e-4, "feature vector")


@pytest.mark.parametrize("name", ["full_tracking", "full_tracking_poisson", "full_tracking_rw", "full_tracking_7scenes"])
def test_model_fused_engine_vs_reference_goldens(name):
    """SLAMNetwork with engine='fused': output dict and the pose gradient of the tracking objective."""
    fx, model, cam, pose, _, _ = _setup(name)
    model.engine = "fused"
    model.draws = draws_of(fx, "cuda")
    model.draws["z_vals_override"] = tt(fx["out_z_vals"]).cuda()
    out = model({"intrinsics": tt(fx["in_K"]).cuda(), "uv": tt(fx["in_uv"]).cuda(), "pose": pose},
                torch.arange(pose.shape[0], device="cuda"), {}, mode="tracking", frame_idx=1)
    assert model.last_engine == "fused"
    assert_outputs_close(out, fx, ("depth_vals", "sdf", "weights", "rgb", "rgb_values", "depth_values", "entropy", "normal_map"))
    loss = golden_objective(out, fx, "tracking")
    assert_close(loss, fx["out_loss"], 1e-6, 1e-5, "loss")
    loss.backward()
    assert_close(cam.grad, fx["grad_cam"], 1e-3 * float(np.abs(fx["grad_cam"]).max()), 1e-3, "grad_cam")


@pytest.mark.parametrize("name,stage,cstage", [("full_tracking", "fine", "highfreq"), ("full_tracking_poisson", "fine", "base"),
                                               ("full_mapping", "coarse", "highfreq"), ("full_tracking_rw", "fine", "highfreq"),
                                               ("full_mapping_rw", "fine", "highfreq"), ("full_tracking_7scenes", "fine", "highfreq"),
                                               ("full_mapping_7scenes", "coarse", "base")])
def test_backward_all_cotangents_vs_oracle(name, stage, cstage):
    """Every differentiable output (rgb, depth, normal map, entropy, weights) pulled back to the pose and compared
    with the CPU oracle (torch autograd over the restated reference graph) -- both stages / colour stages."""
    from oracle import render_ref as R
    fx, model, cam, pose, _, _ = _setup(name)
    model.train(True)
    model.engine = "fused"
    d = draws_of(fx, "cuda")
    d["z_vals_override"] = tt(fx["out_z_vals"]).cuda()
    model.draws = d
    g = torch.Generator().manual_seed(11)
    n_ray = fx["out_z_vals"].shape[0]
    t_rgb, t_dep = torch.rand(n_ray, 3, generator=g), torch.rand(n_ray, 1, generator=g) * 2
    t_nrm = F.normalize(torch.randn(n_ray, 3, generator=g), dim=-1)
    t_w = torch.rand(fx["out_z_vals"].shape, generator=g)

    def objective(out, dev):
        loss = (out["rgb_values"].reshape(-1, 3) - t_rgb.to(dev)).abs().mean()
        loss = loss + 0.3 * (out["depth_values"].reshape(-1, 1) - t_dep.to(dev)).abs().mean()
        n = F.normalize(out["normal_map"].reshape(-1, 3), p=2, dim=-1)
        loss = loss + 0.2 * (n - t_nrm.to(dev)).abs().sum(-1).mean() + 0.05 * out["entropy"]
        return loss + 0.1 * (out["weights"] * t_w.to(dev)).sum(-1).mean()

    out = model({"intrinsics": tt(fx["in_K"]).cuda(), "uv": tt(fx["in_uv"]).cuda(), "pose": pose},
                torch.arange(pose.shape[0], device="cuda"), {}, mode="tracking", stage=stage, color_stage=cstage, frame_idx=1)
    assert model.last_engine == "fused"
    objective(out, "cuda").backward()
    cfg, params = oracle_config(fx), params_of(fx)
    cam_c = tt(fx["in_cam"]).requires_grad_(True)
    dc = draws_of(fx)
    dc["z_vals_override"] = tt(fx["out_z_vals"])
    ref = R.render(params, cfg, tt(fx["in_uv"]), R.camera_from_tensor(cam_c), tt(fx["in_K"]), tt(fx["in_voxels"]), dc,
                   mode="tracking", stage=stage, color_stage=cstage, training=True)
    for k in ("rgb_values", "depth_values", "normal_map", "weights", "entropy"):
        assert_close(out[k], ref[k], 2e-5, 1e-4, k)
    objective(ref, "cpu").backward()
    assert_close(cam.grad, cam_c.grad, 1e-3 * float(cam_c.grad.abs().max()), 1e-3, "grad_cam")


def test_bench_size_fused_vs_composed_engine():
    """BASELINE size (1024 rays x 128 samples, shipped SDF grids, 2^19-capped colour grid to keep the test light):
    the fused engine and the composed engine (torch autograd around the HIP hash operator) agree on the rendered
    colours and on the pose gradient from the same sample positions."""
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.utils.general import get_camera_from_tensor

    class DS:
        img_res = (680, 1200)
    torch.manual_seed(0)
    conf = replica_model_conf(94, 640, 32, use_warp_loss=False)
    model = SLAMNetwork(conf, dataset=DS(), n_images=1,
                        colour_grid=dict(base_resolution=16, desired_resolution=2048, log2_hashmap_size=19)).cuda()
    model.train()
    g = torch.Generator(device="cuda").manual_seed(3)
    for enc, s in ((model.implicit_network.coarse.encoding, 0.02), (model.implicit_network.fine.encoding, 0.02),
                   (model.rendering_network.encoding, 0.3)):
        enc.embeddings.data = (torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1) * s
    for p in model.parameters():
        p.requires_grad_(False)
    R = 1024
    idx = torch.randint(680 * 1200, (1, R), device="cuda", generator=g)
    uv = torch.stack([(idx % 1200).float(), (idx // 1200).float()], -1)
    K = torch.eye(4, device="cuda")
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    gt = torch.rand(R, 3, device="cuda", generator=g)
    t_rand = torch.rand(R, 640, device="cuda", generator=g)
    res = {}
    zfix = None
    for engine in ("fused", "composed"):
        model.engine = engine
        model.draws = {"t_rand": t_rand, "extra_idx": torch.arange(0, 640, 20, device="cuda"),
                       "eik_idx": torch.zeros(R, dtype=torch.long, device="cuda")}
        if zfix is not None:
            model.draws["z_vals_override"] = zfix
        cam = torch.tensor([1.0, 0.01, -0.02, 0.015, 0.1, 0.0, -0.2], device="cuda", requires_grad=True)
        out = model({"intrinsics": K[None], "uv": uv, "pose": get_camera_from_tensor(cam).unsqueeze(0)},
                    torch.zeros(1, dtype=torch.long, device="cuda"), {}, mode="tracking", frame_idx=1)
        zfix = out["z_vals"].detach()
        (out["rgb_values"].reshape(-1, 3) - gt).abs().mean().backward()
        res[engine] = (out["rgb_values"].detach(), cam.grad.clone(), out["weights"].detach(), model.last_engine)
    assert res["fused"][3] == "fused" and res["composed"][3] == "composed"
    assert_close(res["fused"][2], res["composed"][2], 2e-5, 1e-4, "weights")
    assert_close(res["fused"][0], res["composed"][0], 2e-5, 1e-4, "rgb_values")
    assert_close(res["fused"][1], res["composed"][1], 2e-3 * float(res["composed"][1].abs().max()), 2e-3, "grad_cam")


def test_fused_rays_vs_torch_path():
    """nsa_rays_forward / nsa_rays_pose_backward vs the torch restatement of rend_util.get_camera_params (batch of 2
    poses, skewed intrinsics, ragged pixel count)."""
    from nicer_slam_amd.fused import render as fr
    from nicer_slam_amd.utils import rend_util
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    g = torch.Generator().manual_seed(2)
    b, n = 2, 203
    uv = torch.stack([torch.randint(1200, (b, n), generator=g).float(), torch.randint(680, (b, n), generator=g).float()], -1).cuda()
    K = torch.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1] = 600.0, 590.0, 599.5, 339.5, 0.3
    K = K[None].repeat(b, 1, 1).cuda()
    cam = torch.tensor([[1.0, 0.02, -0.01, 0.03, 0.1, 0.0, -0.2], [0.9, -0.1, 0.05, 0.2, -0.3, 0.2, 0.1]]).cuda()
    w_o, w_d = torch.randn(b * n, 3, generator=g).cuda(), torch.randn(b * n, 3, generator=g).cuda()
    res = []
    for fused in (True, False):
        c = cam.clone().requires_grad_(True)
        pose = get_camera_from_tensor(c)
        if fused:
            o, d, ds = fr.rays(pose, uv, K)
        else:
            dd, oo = rend_util.get_camera_params(uv, pose, K)
            d, o = dd.reshape(-1, 3), oo.unsqueeze(1).repeat(1, n, 1).reshape(-1, 3)
            eye = torch.eye(4, device="cuda")[None].repeat(b, 1, 1)
            ds = rend_util.get_camera_params(uv, eye, K)[0][:, :, 2].reshape(-1)
        ((o * w_o).sum() + (d * w_d).sum()).backward()
        res.append((o.detach(), d.detach(), ds.detach(), c.grad.clone()))
    for i, what in enumerate(("rays_o", "rays_d", "depth_scale")):
        assert_close(res[0][i], res[1][i], 1e-7, 2e-6, what)
    assert_close(res[0][3], res[1][3], 1e-4 * float(res[1][3].abs().max()), 1e-4, "camera-tensor gradient")


def test_graph_captured_tracking_step_matches_eager():
    """TrackingStepper: the hipGraph-captured iteration (forward + backward + Adam in one replay) follows exactly the
    same camera trajectory as the eager iteration when both see the same pre-drawn randoms."""
    from nicer_slam_amd.tracking import TrackingStepper
    fx, model, cam, pose, _, _ = _setup("full_tracking")
    model.train(True)
    model.engine = "fused"
    model.draws = draws_of(fx, "cuda")
    K, uv, gt = tt(fx["in_K"]).cuda(), tt(fx["in_uv"]).cuda(), tt(fx["gt_rgb"]).cuda()
    cam0 = tt(fx["in_cam"]).reshape(-1)
    out = {}
    for use_graph in (False, True):
        st = TrackingStepper(model, K, uv.shape[1], cam0, lr=0.005, use_graph=use_graph)
        losses = [float(st.step(uv, gt)) for _ in range(4)]
        out[use_graph] = (st.cam.detach().clone(), losses)
    assert_close(out[True][0], out[False][0], 1e-6, 1e-5, "camera after 4 steps")
    assert abs(out[True][1][0] - float(fx["out_loss"])) < 5e-5      # first-iteration loss = the reference's
    assert out[True][1][-1] < out[True][1][0]                        # and tracking actually descends


def test_kernel_tracker_matches_autograd_stepper():
    """KernelTracker (no autograd: cam->pose, L1, Adam and all backward steps as our kernels) follows the same camera
    trajectory as the autograd-driven TrackingStepper, eagerly and as a hipGraph."""
    from nicer_slam_amd.tracking import TrackingStepper, KernelTracker
    fx, model, cam, pose, _, _ = _setup("full_tracking")
    model.train(True)
    model.engine = "fused"
    model.draws = draws_of(fx, "cuda")
    K, uv, gt = tt(fx["in_K"]).cuda(), tt(fx["in_uv"]).cuda(), tt(fx["gt_rgb"]).cuda()
    cam0 = tt(fx["in_cam"]).reshape(-1)
    ref = TrackingStepper(model, K, uv.shape[1], cam0, lr=0.005, use_graph=False)
    ref_l = [float(ref.step(uv, gt)) for _ in range(5)]
    for use_graph in (False, True):
        kt = KernelTracker(model, K, uv.shape[1], cam0, lr=0.005, use_graph=use_graph)
        ls = [float(kt.step(uv, gt)) for _ in range(5)]
        assert_close(torch.tensor(ls), torch.tensor(ref_l), 2e-6, 1e-5, f"losses (graph={use_graph})")
        assert_close(kt.cam, ref.cam.detach(), 2e-6, 1e-4, f"camera after 5 steps (graph={use_graph})")
    assert abs(ref_l[0] - float(fx["out_loss"])) < 5e-5


@pytest.mark.parametrize("chunks", [1, 2])
def test_kernel_tracker_graph_follows_mlp_updates_between_frames(chunks):
    """A mapping step between two tracked frames changes the MLPs (version bump -> the packed-weight cache misses -> a new
    snapshot, the old one freed).  A captured graph knows only the old snapshot's address: the tracker owns its snapshots and
    re-packs into them, so the replayed graph renders with the CURRENT weights (tracking.py).  Graph vs a fresh
    eager tracker after an optimizer-like in-place update of every MLP and of a table; also the eager chunked path (packs made
    on the launching stream before the fork)."""
    from nicer_slam_amd.tracking import KernelTracker
    fx, model, cam, pose, _, _ = _setup("full_tracking")
    model.train(True)
    model.engine = "fused"
    model.draws = draws_of(fx, "cuda")
    K, uv, gt = tt(fx["in_K"]).cuda(), tt(fx["in_uv"]).cuda(), tt(fx["gt_rgb"]).cuda()
    cam0 = tt(fx["in_cam"]).reshape(-1)
    n = uv.shape[1]
    kt = KernelTracker(model, K, n, cam0, lr=0.005, use_graph=True, chunks=chunks)
    before = [float(kt.step(uv, gt)) for _ in range(2)]
    g = torch.Generator(device="cuda").manual_seed(11)
    with torch.no_grad():                                     # what optimizer.step() of a mapping iteration does: in place
        for name, p in model.named_parameters():
            if ".lin" in name or "embeddings" in name:
                p.add_(0.05 * float(p.abs().mean()) * torch.randn(p.shape, device="cuda", generator=g))
    torch.cuda.empty_cache()                                  # freed snapshots really go away
    junk = [torch.randn(1 << 16, device="cuda") for _ in range(32)]     # ... and their memory gets recycled
    kt.reset(cam0)
    got = [float(kt.step(uv, gt)) for _ in range(3)]
    fresh = KernelTracker(model, K, n, cam0, lr=0.005, use_graph=False, chunks=chunks)
    want = [float(fresh.step(uv, gt)) for _ in range(3)]
    assert abs(want[0] - before[0]) > 1e-4, "the update must change the rendering for this test to mean anything"
    assert_close(torch.tensor(got), torch.tensor(want), 2e-6, 1e-5, "losses after the update (graph vs fresh eager)")
    assert_close(kt.cam, fresh.cam, 2e-6, 1e-4, "camera")
    del junk


def test_kernel_tracker_steplr_and_min_loss_candidate():
    """The reference's per-frame tracking protocol (volsdf_train.py:396-403,425-446): Adam + StepLR, and the frame's result
    is the camera cloned AFTER the step of the arg-min-loss iteration.  KernelTracker (schedule and candidate inside the
    tail kernel; eager and hipGraph) vs torch.optim.Adam + torch StepLR + the same comparison on the autograd stepper; then
    reset() must reproduce a fresh tracker."""
    from nicer_slam_amd.tracking import TrackingStepper, KernelTracker
    fx, model, cam, pose, _, _ = _setup("full_tracking")
    model.train(True)
    model.engine = "fused"
    model.draws = draws_of(fx, "cuda")
    K, uv = tt(fx["in_K"]).cuda(), tt(fx["in_uv"]).cuda()
    g = torch.Generator().manual_seed(3)
    gts = [torch.rand(uv.shape[1], 3, generator=g).cuda() for _ in range(7)]     # a different target per iteration:
    cam0 = tt(fx["in_cam"]).reshape(-1)                                          # the loss is not monotone
    ref = TrackingStepper(model, K, uv.shape[1], cam0, lr=0.01, use_graph=False, lr_step=3, lr_gamma=0.5)
    ref_l, ref_cams = [], []
    for gt in gts:
        ref_l.append(float(ref.step(uv, gt)))
        ref_cams.append(ref.cam.detach().clone())
    best = min(range(len(ref_l)), key=lambda i: (ref_l[i], i))
    assert 0 < best < len(ref_l) - 1 or len(set(ref_l)) > 1
    assert_close(ref.candidate, ref_cams[best], 0, 0, "stepper candidate = camera after the arg-min step")
    for use_graph in (False, True):
        kt = KernelTracker(model, K, uv.shape[1], cam0, lr=0.01, use_graph=use_graph, lr_step=3, lr_gamma=0.5)
        for rep in range(2):                                   # second pass: reset() == a fresh tracker
            ls = [float(kt.step(uv, gt)) for gt in gts]
            assert_close(torch.tensor(ls), torch.tensor(ref_l), 2e-6, 1e-5, f"losses (graph={use_graph}, pass {rep})")
            assert_close(kt.cam, ref_cams[-1], 2e-6, 1e-4, "final camera (StepLR applied)")
            assert_close(kt.candidate, ref_cams[best], 2e-6, 1e-4, "candidate camera")
            assert abs(float(kt.min_loss) - ref_l[best]) < 2e-6
            kt.reset(cam0)
            assert float(kt.min_loss) == 1e10 and float(kt.t) == 0


def test_kernel_tracker_multi_gpu_message_path_single_rank_group():
    """The N > 1 step (tail kernel writes the 9-float message [w*g_cam, w*loss, w]; one all-reduce; Adam on msg/msg[8])
    run over a 1-rank RCCL group must follow exactly the single-GPU trajectory."""
    import socket
    import torch.distributed as dist
    from nicer_slam_amd.tracking import KernelTracker
    fx, model, cam, pose, _, _ = _setup("full_tracking")
    model.train(True)
    model.engine = "fused"
    model.draws = draws_of(fx, "cuda")
    K, uv, gt = tt(fx["in_K"]).cuda(), tt(fx["in_uv"]).cuda(), tt(fx["gt_rgb"]).cuda()
    cam0 = tt(fx["in_cam"]).reshape(-1)
    one = KernelTracker(model, K, uv.shape[1], cam0, lr=0.005, use_graph=True)
    l1 = [float(one.step(uv, gt)) for _ in range(4)]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        many = KernelTracker(model, K, uv.shape[1], cam0, lr=0.005, use_graph=True, world=2)   # forces the message path
        l2 = [float(many.step(uv, gt)) for _ in range(4)]
        assert float(many.red[8]) == float(uv.shape[1])
        # the same with the all-reduce and the Adam step captured into the hipGraph (opt-in)
        cap = KernelTracker(model, K, uv.shape[1], cam0, lr=0.005, use_graph=True, world=2, graph_collective=True)
        assert cap.collective_in_graph
        l3 = [float(cap.step(uv, gt)) for _ in range(4)]
    finally:
        dist.destroy_process_group()
    assert_close(torch.tensor(l2), torch.tensor(l1), 1e-6, 1e-5, "losses")
    assert_close(many.cam, one.cam, 1e-6, 1e-5, "camera after 4 steps")
    assert_close(torch.tensor(l3), torch.tensor(l1), 1e-6, 1e-5, "losses, captured collective")
    assert_close(cap.cam, one.cam, 1e-6, 1e-5, "camera after 4 steps, captured collective")


@pytest.mark.parametrize("chunks", [2, 3])
def test_kernel_tracker_chunked_streams_follow_the_single_stream_trajectory(chunks):
    """chunks > 1: independent ray chunks on their own streams (fork / join inside the captured graph), joined through the
    weighted 9-float message -- same arithmetic per ray, so with pinned draws the trajectory is the single-stream one."""
    from nicer_slam_amd.tracking import KernelTracker
    fx, model, cam, pose, _, _ = _setup("full_tracking")
    model.train(True)
    model.engine = "fused"
    model.draws = draws_of(fx, "cuda")
    K, uv, gt = tt(fx["in_K"]).cuda(), tt(fx["in_uv"]).cuda(), tt(fx["gt_rgb"]).cuda()
    cam0 = tt(fx["in_cam"]).reshape(-1)
    one = KernelTracker(model, K, uv.shape[1], cam0, lr=0.005, use_graph=True)
    l1 = [float(one.step(uv, gt)) for _ in range(4)]
    for use_graph in (False, True):
        many = KernelTracker(model, K, uv.shape[1], cam0, lr=0.005, use_graph=use_graph, chunks=chunks)
        l2 = [float(many.step(uv, gt)) for _ in range(4)]
        assert float(many.red[8]) == float(uv.shape[1])
        assert_close(torch.tensor(l2), torch.tensor(l1), 1e-6, 1e-5, "losses")
        assert_close(many.cam, one.cam, 1e-6, 1e-5, "camera after 4 steps")
        assert_close(many.candidate, one.candidate, 1e-6, 1e-5, "arg-min-loss camera")


def test_composite_backward_far_from_the_surface_matches_torch_expm1_backward():
    """Regression (found by tests/test_configs_gpu.py): in empty space (sdf / beta > ~17) expm1(-|s|/beta) has rounded to
    exactly -1, so the reference's sigma is exactly 0 or one ulp of 0.5 / beta, and torch's expm1 backward -- (result + 1),
    not exp(x) -- gives a derivative that is exactly 0 or 2^-24-quantised.  The last interval multiplies d sigma by 1e10, so
    an analytic exp(-|s|/beta) there (1e-8 .. 1e-16, not 0) shows up as a spurious gradient of ordinary magnitude.
    Synthetic rays through empty space, per-sample d/d sdf vs torch autograd over the oracle's volume_weights."""
    from oracle import render_ref as R
    from nicer_slam_amd._native import lib, check
    n_ray, S = 64, 64
    g = torch.Generator().manual_seed(2)
    z = torch.sort(torch.rand(n_ray, S, generator=g) * 1.5, dim=1).values
    ro = torch.tensor([0.1, 0.0, -0.2]).repeat(n_ray, 1)
    rd = torch.nn.functional.normalize(torch.randn(n_ray, 3, generator=g), dim=-1) * 0.5
    # sdf from just below the surface to deep in empty space: |s| / beta spans 0 .. 70 (beta = 0.01444 at zero visits)
    sdf0 = (torch.rand(n_ray, S, generator=g) * 1.0 - 0.02)
    sdf0[:, -1] = torch.linspace(0.05, 0.9, n_ray)                 # the 1e10 interval, at every depth of the quantised zone
    rgb = torch.rand(n_ray, S, 3, generator=g)
    grad = torch.randn(n_ray, S, 3, generator=g)
    vox = torch.zeros(64, 64, 64)
    g_rgbv = torch.rand(n_ray, 3, generator=g) - 0.5
    dev = lambda t: t.cuda().contiguous()
    zc, roc, rdc, sc, rc, gc, vc, gr = map(dev, (z, ro, rd, sdf0, rgb, grad, vox, g_rgbv))
    P = n_ray * S
    g_sdf, g_rgb, g_grad = (torch.empty(P, device="cuda"), torch.empty(P, 3, device="cuda"), torch.empty(P, 3, device="cuda"))
    check(lib.nsa_composite_backward(roc.data_ptr(), rdc.data_ptr(), zc.data_ptr(), sc.data_ptr(), rc.data_ptr(), gc.data_ptr(),
                                     vc.data_ptr(), 64, n_ray, S, gr.data_ptr(), None, None, None, None, g_sdf.data_ptr(),
                                     g_rgb.data_ptr(), g_grad.data_ptr(), torch.cuda.current_stream().cuda_stream))
    sdf = sdf0.reshape(-1, 1).clone().requires_grad_(True)
    pts = (ro.unsqueeze(1) + z.unsqueeze(2) * rd.unsqueeze(1)).reshape(-1, 3)
    w = R.volume_weights(z, sdf, pts, vox, 64)
    ((w.unsqueeze(-1) * rgb).sum(1) * g_rgbv).sum().backward()
    ref = sdf.grad.reshape(n_ray, S)
    got = g_sdf.cpu().reshape(n_ray, S)
    # the reference's sigma is quantised: either exactly 0 (derivative 0) or >= 2e-6, which saturates the 1e10 interval
    # (exp(-2e4) = 0): d/d sdf of the last sample is exactly zero on every ray, whatever its depth
    assert bool((ref[:, -1] == 0).all())
    assert float(got[:, -1].abs().max()) == 0.0
    assert_close(got, ref, 1e-7 + 2e-3 * float(ref.abs().max()), 2e-3, "d/d sdf")


def test_composite_backward_exact_zero_on_saturated_last_interval():
    """Regression: the last interval is 1e10 long, so d(weights)/d(sdf_last) is exactly 0 whenever sigma_last * 1e10
    saturates alpha (the usual case) -- a 'total minus prefix' suffix sum leaks a rounding residue times 1e10 there.
    Compared per sample with torch autograd over the oracle's volume_weights."""
    import ctypes
    from oracle import render_ref as R
    from nicer_slam_amd.fused import render as fr
    from nicer_slam_amd._native import lib, check
    fx, model, cam, pose, rays_o, rays_d = _setup("full_tracking_poisson")
    z = tt(fx["out_z_vals"]).cuda()
    ro, rd = rays_o.detach().contiguous(), rays_d.detach().contiguous()
    b = fr.composite_forward_raw(model, ro, rd, z, "fine", True)
    n_ray, S = z.shape
    g = torch.Generator().manual_seed(0)
    g_rgbv = (torch.rand(n_ray, 3, generator=g) - 0.5).cuda().contiguous()
    P = n_ray * S
    g_sdf, g_rgb, g_grad = (torch.empty(P, device="cuda"), torch.empty(P, 3, device="cuda"), torch.empty(P, 3, device="cuda"))
    check(lib.nsa_composite_backward(ro.data_ptr(), rd.data_ptr(), z.data_ptr(), b["sdf"].data_ptr(), b["rgb"].data_ptr(),
                                     b["grad"].data_ptr(), b["vox"].data_ptr(), 64, n_ray, S, g_rgbv.data_ptr(), None, None,
                                     None, None, g_sdf.data_ptr(), g_rgb.data_ptr(), g_grad.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream))
    sdf = b["sdf"].detach().cpu().reshape(-1, 1).requires_grad_(True)
    pts = (ro.cpu().unsqueeze(1) + z.cpu().unsqueeze(2) * rd.cpu().unsqueeze(1)).reshape(-1, 3)
    w = R.volume_weights(z.cpu(), sdf, pts, tt(fx["in_voxels"]), 64)
    rgbv = (w.unsqueeze(-1) * b["rgb"].detach().cpu().reshape(n_ray, S, 3)).sum(1)
    (rgbv * g_rgbv.cpu()).sum().backward()
    ref = sdf.grad.reshape(n_ray, S)
    got = g_sdf.cpu().reshape(n_ray, S)
    assert_close(got, ref, 3e-6, 2e-3, "d/d sdf")    # values are O(1e-6..1e-4): differences of nearly cancelling terms
    sat = ref[:, -1] == 0
    assert bool(sat.any())
    assert bool((got[:, -1][sat] == 0).all())


@pytest.mark.parametrize("n_rays,samples", [(37, (5, 24, 3)), (3, (2, 9, 0)), (130, (60, 70, 4)), (96, (158, 640, 32)),
                                            (20, (222, 640, 32))])
def test_ragged_sizes_fused_vs_composed(n_rays, samples):
    """Ray counts that are not multiples of 4 / point counts that are not multiples of 32 / no extra samples /
    S > 64 (two samples per lane in the per-ray kernels) / S = 192 (BASELINE configs[4]: 8192 rays x 192 samples, three
    samples per lane, 256-key sort) / S = 256 (the supported maximum): fused vs composed engine, values and pose gradient."""
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.utils.general import get_camera_from_tensor

    class DS:
        img_res = (680, 1200)
    torch.manual_seed(1)
    conf = replica_model_conf(*samples, use_warp_loss=False)
    conf["implicit_network"]["fine"].update(end_size=64, logmap=12)
    model = SLAMNetwork(conf, dataset=DS(), n_images=1,
                        colour_grid=dict(base_resolution=16, desired_resolution=128, log2_hashmap_size=12)).cuda()
    model.train()
    g = torch.Generator(device="cuda").manual_seed(5)
    for enc, s in ((model.implicit_network.coarse.encoding, 0.03), (model.implicit_network.fine.encoding, 0.03),
                   (model.rendering_network.encoding, 0.4)):
        enc.embeddings.data = (torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1) * s
    for p in model.parameters():
        p.requires_grad_(False)
    E = samples[1]
    idx = torch.randint(680 * 1200, (1, n_rays), device="cuda", generator=g)
    uv = torch.stack([(idx % 1200).float(), (idx // 1200).float()], -1)
    K = torch.eye(4, device="cuda")
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    gt = torch.rand(n_rays, 3, device="cuda", generator=g)
    S = samples[0] + 2 + samples[2]
    draws = {"t_rand": torch.rand(n_rays, E, device="cuda", generator=g),
             "extra_idx": torch.randperm(E, device="cuda", generator=g)[:samples[2]],
             "eik_idx": torch.randint(S, (n_rays,), device="cuda", generator=g)}
    res, zfix = {}, None
    for engine in ("fused", "composed"):
        model.engine = engine
        model.draws = dict(draws)
        if zfix is not None:
            model.draws["z_vals_override"] = zfix
        cam = torch.tensor([1.0, 0.03, -0.02, 0.01, 0.05, 0.02, -0.1], device="cuda", requires_grad=True)
        out = model({"intrinsics": K[None], "uv": uv, "pose": get_camera_from_tensor(cam).unsqueeze(0)},
                    torch.zeros(1, dtype=torch.long, device="cuda"), {}, mode="tracking", frame_idx=1)
        assert out["z_vals"].shape == (n_rays, S)
        zfix = out["z_vals"].detach()
        loss = (out["rgb_values"].reshape(-1, 3) - gt).abs().mean() + 0.1 * out["depth_values"].mean()
        loss.backward()
        res[engine] = (out["rgb_values"].detach(), out["depth_values"].detach(), out["normal_map"].detach(), cam.grad.clone())
    assert model.last_engine == "composed"
    for i, what in enumerate(("rgb_values", "depth_values", "normal_map")):
        assert_close(res["fused"][i], res["composed"][i], 2e-5, 1e-4, what)
    assert_close(res["fused"][3], res["composed"][3], 2e-3 * float(res["composed"][3].abs().max()), 2e-3, "grad_cam")


def test_morton_launch_order_is_transparent():
    """nsa_points_t.order: the per-point kernels run in Morton order, every result is the same as in ray order
    (per-point arithmetic is order independent; the per-ray composite reads per-point arrays)."""
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.fused import render as fr, sampler as fs
    torch.manual_seed(7)
    model = SLAMNetwork(replica_model_conf(use_warp_loss=False)).cuda().train()
    with torch.no_grad():
        for enc in (model.implicit_network.coarse.encoding, model.implicit_network.fine.encoding,
                    model.rendering_network.encoding):
            enc.embeddings.uniform_(-0.05, 0.05)
    R = 200
    o = torch.tensor([0.1, 0.0, -0.2], device="cuda").repeat(R, 1).contiguous()
    d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=-1).contiguous()
    z, _ = fs.get_z_vals(model, d, o, need_eik=False)
    g = torch.rand(R, 3, device="cuda")
    res = []
    for sort_points in (False, True):
        b = fr.composite_forward_raw(model, o, d, z, "fine", True, sort_points=sort_points)
        g_o, g_d = fr.composite_backward_raw(model, o, d, z, b, "fine", "highfreq", g_rgbv=g)
        res.append((b["rgb_values"].clone(), b["depth"].clone(), b["nmap"].clone(), b["sdf"].clone(), b["rgb"].clone(),
                    g_o.clone(), g_d.clone()))
        if sort_points:
            order = b["order"].long()
            assert sorted(order.tolist()) == list(range(R * z.shape[1]))
    for a, c in zip(*res):
        assert torch.equal(a, c)


def test_fused_kernels_reject_more_than_256_samples_per_ray():
    """S > 256 is outside the per-ray kernels' compiled range: the C ABI returns an error code (no silent truncation,
    no fallback) and the binding raises."""
    from nicer_slam_amd.fused import render as fr
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.model.network import SLAMNetwork
    model = SLAMNetwork(replica_model_conf(use_warp_loss=False)).cuda()
    R, S = 8, 300
    o = torch.zeros(R, 3, device="cuda")
    d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=-1)
    z = torch.sort(torch.rand(R, S, device="cuda"), dim=1).values
    with pytest.raises(RuntimeError):
        fr.composite_forward_raw(model, o, d, z, "fine", False)
