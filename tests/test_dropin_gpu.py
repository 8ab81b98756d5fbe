"""B2 drop-in check: the reference's tracking + mapping loops (code/training/volsdf_train.py:393-446 and :451-576), restated
call for call around a model built EXACTLY as the reference builds it (no freeze_fine_mlp(), no requires_grad_ edits, default
engine), with the reference's optimizer groups (torch.optim.Adam, :150-174), its two SLAMLoss instances (confs: loss /
tracking_loss) with the shipped weights incl. patch warp and flow (use_warp_loss = true, mapping_patchsizes = [1]), flow edges between
the keyframes, bundle adjustment in the last 30 % of a mapping round, StepLR(50, 0.95) on the camera and the arg-min-loss candidate.
Every forward must run on the fused engine (the re-projection blocks on the kernels of C ABI section 5)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

H, W = 68, 120


class _DS:
    img_res = (H, W)
    data_dir = "synthetic"


def _world(n_samples=64):
    from nicer_slam_amd.feed import FrameFeed
    from nicer_slam_amd.model.loss import SLAMLoss
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    torch.manual_seed(0)
    model = SLAMNetwork(conf=replica_model_conf(n_samples, use_warp_loss=True, mapping_patchsizes=[1]), dataset=_DS(), n_images=3,   # the shipped confs
                        colour_grid=dict(base_resolution=16, desired_resolution=256, log2_hashmap_size=15))
    model.train_dataset, model.keyframe_every = None, 10
    model.cuda()
    lr = 0.002
    para_list = [     # volsdf_train.py:150-173
        {"name": "encoding", "params": list(model.implicit_network.fine.grid_parameters()), "lr": lr * 20.0},
        {"name": "encoding", "params": list(model.implicit_network.coarse.grid_parameters()), "lr": lr * 20.0},
        {"name": "net", "params": list(model.rendering_network.grid_parameters()), "lr": lr * 5.0},
        {"name": "net", "params": list(model.rendering_network.mlp_parameters()), "lr": lr},
        {"name": "density", "params": list(model.density.parameters()), "lr": 2e-3},
        {"name": "coarse_mlp_parameters", "params": list(model.implicit_network.coarse.mlp_parameters()), "lr": lr},
    ]
    optimizer = torch.optim.Adam(para_list, betas=(0.9, 0.99), eps=1e-15)
    loss = SLAMLoss(model=model, rgb_loss="torch.nn.L1Loss", assign_scale_shift_init=True, eikonal_weight=0.1,     # confs/replica/
                    smooth_weight=0.005, depth_weight=0.1, normal_l1_weight=0.05, normal_cos_weight=0.05,          # runconf_replica_1.conf:45-57
                    warp_loss_weight=0.5, warp_loss_type="l1", flow_weight=0.001)
    tracking_loss = SLAMLoss(model=model, rgb_loss="torch.nn.L1Loss", eikonal_weight=0, smooth_weight=0, depth_weight=0,
                             normal_l1_weight=0, normal_cos_weight=0)
    feed = FrameFeed((H, W), device="cuda")
    g = torch.Generator().manual_seed(1)
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = 60.0
    K[0, 2], K[1, 2] = W / 2 - 0.5, H / 2 - 0.5
    for idx in range(2):
        pose = torch.eye(4)
        pose[:3, 3] = torch.tensor([0.1 + 0.01 * idx, 0.0, -0.2])
        feed.add_frame(idx, rgb=torch.rand(H * W, 3, generator=g), depth=torch.rand(H * W, 1, generator=g) + 0.5,
                       normal=torch.nn.functional.normalize(torch.randn(H * W, 3, generator=g), dim=-1),
                       gt_depth=1.0 + 0.3 * torch.rand(H * W, 1, generator=g), intrinsics=K, pose=pose)
    return model, optimizer, loss, tracking_loss, feed


def test_reference_loops_run_on_the_fused_engine_unmodified():
    from nicer_slam_amd.utils.general import get_camera_from_tensor, get_tensor_from_camera
    model, optimizer, loss_fn, tracking_loss, feed = _world()
    assert all(p.requires_grad for p in model.parameters()) and model.engine == "auto"
    engines = []
    before = {n: p.detach().clone() for n, p in model.named_parameters()}

    seen = {"warp": 0, "flow": 0, "ba": 0}

    def mapping(frame_idx, keyframe_list, iters):
        out_losses = []
        edges = None
        if len(keyframe_list) >= 2:                                  # build_graph (:312-324) for two neighbouring keyframes
            edges = (torch.tensor([0, 1], device="cuda"), torch.tensor([1, 0], device="cuda"), None, None)
            flow_img = torch.randn(2, H * W, 2, device="cuda") * 2
            flow_occ = torch.rand(2, H * W, device="cuda") > 0.2
        for it in range(iters):                                      # :451-576
            sel = feed.change_sampling_idx(512 // len(keyframe_list))
            # frames handed over as resident stores on even iterations, as the reference's stacked tensors on odd ones
            indices, model_input, ground_truth = feed.batch(keyframe_list, full="store" if it % 2 == 0 else "stack")
            ba = frame_idx != 0 and it > int(iters * 0.7)            # :455, :521-528: camera tensors join the optimisation
            if ba:
                cams = torch.stack([get_tensor_from_camera(feed.frames[k]["pose"].cpu()) for k in keyframe_list])
                cams = cams.cuda().requires_grad_(True)
                opt_ba = torch.optim.Adam([cams], lr=0.001)
                model_input["pose"] = get_camera_from_tensor(cams)
                opt_ba.zero_grad()
                seen["ba"] += 1
            if edges is not None:                                    # :541-546 (select_flow_uv)
                ground_truth["edges"] = edges
                ground_truth["flow"], ground_truth["flow_mask"] = flow_img.index_select(1, sel), flow_occ.index_select(1, sel)
            optimizer.zero_grad()
            if frame_idx > 1:
                stage = "coarse" if it < int(iters * 0.25) else "fine"
                color_stage = "base" if it < int(iters * 0.7) else "highfreq"
            else:
                stage, color_stage = "fine", "highfreq"
            out = model(model_input, indices, ground_truth, keyframe_list=keyframe_list, frame_idx=frame_idx, mode="mapping",
                        stage=stage, color_stage=color_stage, iter=it)
            engines.append(("mapping", stage, color_stage, model.last_engine))
            assert "warp_output" in out and set(out["warp_output"]) == {1}
            terms = loss_fn(out, ground_truth, keyframe_list, frame_idx=frame_idx, stage=stage)
            l = terms["loss"]
            seen["warp"] += int(float(terms["warp_loss"]) > 0)
            seen["flow"] += int("flow" in out and float(terms["flow_loss"]) > 0)
            l.backward()
            optimizer.step()
            if ba:
                assert cams.grad is not None and bool(torch.isfinite(cams.grad).all()) and float(cams.grad.abs().max()) > 0
                opt_ba.step()
            out_losses.append(float(l))
        return out_losses

    m0 = mapping(0, [0], 4)
    # tracking of frame 1                                           :393-446
    cam = get_tensor_from_camera(feed.frames[0]["pose"].cpu()).cuda().requires_grad_(True)
    opt_cam = torch.optim.Adam([cam], lr=0.001)
    sched = torch.optim.lr_scheduler.StepLR(opt_cam, step_size=50, gamma=0.95)
    candidate, min_loss = None, 1e10
    feed.change_sampling_idx(256)
    for it in range(5):
        c2w = get_camera_from_tensor(cam)
        indices, model_input, ground_truth = feed.batch([1])
        model_input["pose"] = c2w.unsqueeze(0)
        out = model(model_input, indices, ground_truth, mode="tracking", frame_idx=1)
        engines.append(("tracking", "fine", "highfreq", model.last_engine))
        l = tracking_loss(out, ground_truth, stage="fine", frame_idx=1)["loss"]
        l.backward()
        opt_cam.step()
        sched.step()
        opt_cam.zero_grad()
        if l < min_loss:
            min_loss, candidate = l, cam.clone().detach()
    assert candidate is not None and bool(torch.isfinite(candidate).all())
    feed.set_pose(1, get_camera_from_tensor(candidate).detach())
    m1 = mapping(2, [0, 1], 8)                                       # frame_idx > 1: coarse->fine, base->highfreq schedule
    assert all(e[-1] == "fused" for e in engines), [e for e in engines if e[-1] != "fused"]
    assert {e[1] for e in engines} == {"coarse", "fine"} and {e[2] for e in engines} == {"base", "highfreq"}
    assert all(l == l and abs(l) < 1e6 for l in m0 + m1)
    assert seen["warp"] >= 4 and seen["flow"] >= 6 and seen["ba"] >= 2, seen      # the shipped terms were really in the objective
    moved = {n for n, p in model.named_parameters() if not torch.equal(p.detach(), before[n])}
    assert any("rendering_network.lin" in n for n in moved) and any("coarse.lin" in n for n in moved)
    assert any(n.endswith("encoding.embeddings") for n in moved)
    for n, p in model.named_parameters():                            # the pretrained fine MLP is never stepped (:140-173)
        if n.startswith("implicit_network.fine.lin"):
            assert n not in moved and p.grad is None


# ---------------------------------------------------------------------------------------------------------------------------------
# The tracking forward / backward of the unmodified loop as two cached hipGraphs (fused/track_graph.py)
def _track_loop(model, feed, iters, objective="rgb", bump_at=None, graph=True):
    """volsdf_train.py:406-443 on frame 1 with fresh draw state and seed; returns per-iteration (loss, camera gradient)."""
    from nicer_slam_amd.fused import track_graph
    from nicer_slam_amd.utils.general import get_camera_from_tensor, get_tensor_from_camera
    for k in ("_draw_state", "_draw_states", "_draw_seed", "_track_graphs"):
        model.__dict__.pop(k, None)
    torch.manual_seed(3)
    old = track_graph.ENABLED
    track_graph.ENABLED = graph
    try:
        cam = get_tensor_from_camera(feed.frames[0]["pose"].cpu()).cuda().requires_grad_(True)
        opt_cam = torch.optim.Adam([cam], lr=0.001)
        feed.change_sampling_idx(256, generator=torch.Generator(device="cuda").manual_seed(9))
        out_l, out_g = [], []
        for it in range(iters):
            if bump_at is not None and it == bump_at:            # a mapping step between two tracked frames moved the MLPs
                with torch.no_grad():
                    for n_, p in model.named_parameters():
                        if n_.endswith("weight_v") or n_.endswith("bias"):
                            p.mul_(1.01)
            c2w = get_camera_from_tensor(cam)
            indices, model_input, ground_truth = feed.batch([1])
            model_input["pose"] = c2w.unsqueeze(0)
            out = model(model_input, indices, ground_truth, mode="tracking", frame_idx=1)
            assert model.last_engine == "fused"
            l = (out["rgb_values"].reshape(-1, 3) - ground_truth["rgb"].reshape(-1, 3).cuda()).abs().mean()
            if objective == "all":                               # cotangents on every differentiable output (eager backward body)
                l = l + 0.1 * out["depth_values"].mean() + 0.01 * out["normal_map"].sum() + 0.05 * out["entropy"] + \
                    0.02 * (out["weights"] ** 2).mean()
            l.backward()
            out_l.append(l.detach().clone())
            out_g.append(cam.grad.detach().clone())
            opt_cam.step()
            opt_cam.zero_grad()
        torch.cuda.synchronize()
        tg = model.__dict__.get("_track_graphs", {}).get("tg")
        return torch.stack(out_l), torch.stack(out_g), tg
    finally:
        track_graph.ENABLED = old


@pytest.mark.parametrize("objective", ["rgb", "all"])
def test_graph_cached_tracking_is_the_eager_functions(objective):
    """Same kernels, same order, same draws (the engine's own Philox stream, reseeded): the cached-graph forward / backward must
    return what the eager autograd.Functions return -- the loss AND the camera gradient of every iteration bit for bit (every sum
    of the path is fixed-order since nsa_rays_pose_backward stopped using atomics; with them the gradients differed in their last
    bits and, through the optimizer step, so did every later loss) -- incl. after an MLP update between iterations (snapshots
    re-packed in place)."""
    model, optimizer, loss_fn, tracking_loss, feed = _world()
    start = {k: v.detach().clone() for k, v in model.state_dict().items()}
    la, ga, tg = _track_loop(model, feed, 7, objective, bump_at=4, graph=True)
    assert tg is not None and tg.fwd_graph is not None and tg.bwd_graph is not None and tg.calls == 7
    with torch.no_grad():                                          # (the loop moved the MLPs at iteration 4: same start for both)
        for k, v in model.state_dict().items():
            v.copy_(start[k])
    lb, gb, none = _track_loop(model, feed, 7, objective, bump_at=4, graph=False)
    assert none is None
    assert torch.equal(la, lb), (la - lb).abs().max()
    assert torch.equal(ga, gb), ((ga - gb).abs().max(), gb.abs().max())
    assert bool(torch.isfinite(ga).all()) and float(ga.abs().max()) > 0
    assert not torch.equal(ga[3], ga[4])


def test_graph_cached_tracking_refuses_a_backward_through_overwritten_outputs():
    from nicer_slam_amd.utils.general import get_camera_from_tensor, get_tensor_from_camera
    model, optimizer, loss_fn, tracking_loss, feed = _world()
    cam = get_tensor_from_camera(feed.frames[0]["pose"].cpu()).cuda().requires_grad_(True)
    feed.change_sampling_idx(256)
    outs = []
    for _ in range(2):
        indices, model_input, ground_truth = feed.batch([1])
        model_input["pose"] = get_camera_from_tensor(cam).unsqueeze(0)
        outs.append(model(model_input, indices, ground_truth, mode="tracking", frame_idx=1))
    outs[1]["rgb_values"].sum().backward()                        # the latest forward: fine
    with pytest.raises(RuntimeError, match="EARLIER forward"):
        outs[0]["rgb_values"].sum().backward()


def test_graph_cached_tracking_outputs_survive_the_next_iterations():
    """ADVICE r4 (medium) / VERDICT r4 weak #8: a caller that KEEPS model_outputs across iterations -- the best-iteration render,
    a logged PSNR, detached outputs (volsdf_train.py:417-446 keeps `model_outputs` of the last iteration around) -- must read
    that iteration's values, as with the eager path: the per-ray results are fresh tensors; the per-sample tensors are documented
    views of the graph's static buffers (fused/track_graph.py) unless NSA_TRACK_CLONE=all."""
    from nicer_slam_amd.fused import track_graph
    from nicer_slam_amd.utils.general import get_camera_from_tensor, get_tensor_from_camera
    model, optimizer, loss_fn, tracking_loss, feed = _world()
    cam = get_tensor_from_camera(feed.frames[0]["pose"].cpu()).cuda().requires_grad_(True)
    opt_cam = torch.optim.Adam([cam], lr=0.01)
    feed.change_sampling_idx(256, generator=torch.Generator(device="cuda").manual_seed(9))
    small = ("rgb_values", "depth_values", "normal_map", "entropy")
    kept, snap = [], []
    for it in range(5):                                           # iteration 0 eager warm-up, 1 captures, 2.. replay
        indices, model_input, ground_truth = feed.batch([1])
        model_input["pose"] = get_camera_from_tensor(cam).unsqueeze(0)
        out = model(model_input, indices, ground_truth, mode="tracking", frame_idx=1)
        kept.append(out)                                          # held, not cloned
        snap.append({k: out[k].detach().clone() for k in small + ("weights",)})
        (out["rgb_values"].reshape(-1, 3) - ground_truth["rgb"].reshape(-1, 3).cuda()).abs().mean().backward()
        opt_cam.step()
        opt_cam.zero_grad()
    torch.cuda.synchronize()
    tg = model.__dict__["_track_graphs"]["tg"]
    assert tg.fwd_graph is not None
    for it in range(5):
        for k in small:
            assert torch.equal(kept[it][k].detach(), snap[it][k]), (it, k)
    assert not torch.equal(snap[2]["rgb_values"], snap[3]["rgb_values"])          # the iterations really differ
    # the documented exception: per-sample tensors of replayed iterations alias the static buffers ...
    assert kept[2]["weights"].data_ptr() == kept[3]["weights"].data_ptr()
    assert torch.equal(kept[2]["weights"].detach(), snap[4]["weights"])
    # ... unless the caller asks for clones of everything
    old = track_graph.CLONE
    track_graph.CLONE = "all"
    try:
        outs = []
        for it in range(2):
            indices, model_input, ground_truth = feed.batch([1])
            model_input["pose"] = get_camera_from_tensor(cam).unsqueeze(0)
            outs.append(model(model_input, indices, ground_truth, mode="tracking", frame_idx=1))
            snap.append(outs[-1]["weights"].detach().clone())
        assert outs[0]["weights"].data_ptr() != outs[1]["weights"].data_ptr()
        assert torch.equal(outs[0]["weights"].detach(), snap[-2])
    finally:
        track_graph.CLONE = old


@pytest.mark.parametrize("n_samples", [64, 94])      # S = 98 (shipped confs: nsa_composite_track) and S = 128 (the bench shape: nsa_colour_forward_track)
def test_tracking_objective_through_the_loss_class_seam(n_samples):
    """the reference resolves its losses from the conf string `train.loss_class` (volsdf_train.py:117-130) and hands the
    loader's ground-truth dict to the model AND to `tracking_loss` (:415-424).  With the class string pointed at this package, the cached
    tracking forward folds the L1 term in (out["tracking_rgb_l1"]) and the loss class returns it -- same value as torch's L1Loss on
    rgb_values, same pose gradient as the backward through rgb_values (both taken from ONE forward: the engine's own draws differ per call),
    and the loop of :406-446 runs on it.  A loss that is NOT the tracking configuration, or another ground-truth tensor, takes the ordinary path."""
    from nicer_slam_amd.utils.general import get_class, get_camera_from_tensor, get_tensor_from_camera
    model, optimizer, loss_fn, _, feed = _world(n_samples)
    conf = {"train": {"loss_class": "nicer_slam_amd.model.loss.SLAMLoss"},                      # the one-line swap
            "tracking_loss": dict(rgb_loss="torch.nn.L1Loss", eikonal_weight=0, smooth_weight=0, depth_weight=0, normal_l1_weight=0,
                                  normal_cos_weight=0)}                                          # confs/replica/runconf_replica_1.conf:58-65
    tracking_loss = get_class(conf["train"]["loss_class"])(trainer=None, train_dataset=None, scan_id=1, model=model, **conf["tracking_loss"])
    model.tracking_param_grads = False
    cam = get_tensor_from_camera(feed.frames[0]["pose"].cpu()).cuda().requires_grad_(True)
    opt_cam = torch.optim.Adam([cam], lr=0.001)
    feed.change_sampling_idx(256)
    losses = []
    for it in range(6):                       # (call 1 eager, call 2 captures the graphs, calls 3.. replay them)
        indices, model_input, ground_truth = feed.batch([1])
        model_input["pose"] = get_camera_from_tensor(cam).unsqueeze(0)
        out = model(model_input, indices, ground_truth, mode="tracking", frame_idx=1)
        assert model.last_engine == "fused" and "tracking_rgb_l1" in out and out["tracking_rgb_l1"][1] is ground_truth["rgb"]
        assert out["z_vals"].shape[1] == n_samples + 34
        terms = tracking_loss(out, ground_truth, stage="fine", frame_idx=1)
        l = terms["loss"]
        assert l is out["tracking_rgb_l1"][0] and terms["rgb_loss"] is l and terms["depth_loss"] == 0.0
        ref = (out["rgb_values"].reshape(-1, 3) - ground_truth["rgb"].reshape(-1, 3)).abs().mean()
        assert abs(float(l) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref))), (float(l), float(ref))
        (g_seam,) = torch.autograd.grad(l, cam, retain_graph=True)
        (g_ref,) = torch.autograd.grad(ref, cam, retain_graph=True)
        assert float(g_ref.abs().max()) > 0
        assert float((g_seam - g_ref).abs().max()) <= 2e-5 * float(g_ref.abs().max()), (it, g_seam, g_ref)
        # the full SLAMLoss (mapping weights) must NOT take the shortcut, nor the tracking loss on another ground-truth tensor
        assert loss_fn._tracking_objective(out, ground_truth) is None
        other = dict(ground_truth, rgb=ground_truth["rgb"].clone())
        l_other = tracking_loss(out, other, stage="fine", frame_idx=1)["loss"]
        assert l_other is not l and abs(float(l_other) - float(ref)) <= 1e-6
        (0.5 * l + 0.5 * l_other).backward()              # objective cotangent beside an rgb_values cotangent: the mixed path
        assert float((cam.grad - g_ref).abs().max()) <= 2e-5 * float(g_ref.abs().max())
        opt_cam.step()
        opt_cam.zero_grad()
        losses.append(float(l))
    assert all(v == v for v in losses)
    # a [R,3]-shaped or host ground truth is accepted too (the reference's own loader hands host tensors over)
    indices, model_input, ground_truth = feed.batch([1])
    model_input["pose"] = get_camera_from_tensor(cam).unsqueeze(0)
    host = dict(ground_truth, rgb=ground_truth["rgb"].cpu())
    out = model(model_input, indices, host, mode="tracking", frame_idx=1)
    l = tracking_loss(out, host, stage="fine", frame_idx=1)["loss"]
    ref = (out["rgb_values"].reshape(-1, 3) - ground_truth["rgb"].reshape(-1, 3)).abs().mean()
    assert l is out["tracking_rgb_l1"][0] and abs(float(l) - float(ref)) <= 1e-6
