"""The folded tracking sequence (nsa_track_begin, nsa_composite_track, nsa_track_finish: C ABI section 3) against the plain
entry points it replaces -- the same arithmetic per ray, so the per-sample cotangents must be bit-identical; the ray sums are
added in a different (fixed) order, so the camera gradient and the loss agree to rounding and are reproducible bit for bit.
Reference: code/model/network.py:349-370, loss.py:57-65,131, training/volsdf_train.py:406-446.  Needs an MI355X."""
import os

import pytest
import torch

from helpers import load, tt, draws_of, assert_close
from test_fused_gpu import _setup

pytestmark = pytest.mark.gpu


def _st():
    return torch.cuda.current_stream().cuda_stream


def _ray_batch(R, S, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.rand(*s, device="cuda", generator=g)
    rays_o = (r(1, 3) * 0.2 - 0.1).repeat(R, 1).contiguous()
    rays_d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda", generator=g), dim=-1)
    z = torch.sort(r(R, S) * 1.5 + 0.05, dim=1).values.contiguous()
    sdf = (torch.randn(R, S, device="cuda", generator=g) * 0.05).contiguous()
    rgb = r(R, S, 3).contiguous()
    grad = torch.randn(R, S, 3, device="cuda", generator=g).contiguous()
    vox = (r(32, 32, 32) * 50).floor().contiguous()
    gt = r(R, 3).contiguous()
    return rays_o, rays_d, z, sdf, rgb, grad, vox, gt


@pytest.mark.parametrize("R,S", [(1024, 128), (37, 98), (5, 200)])
def test_composite_track_is_forward_l1_backward_in_one_pass(R, S):
    from nicer_slam_amd._native import lib, check
    rays_o, rays_d, z, sdf, rgb, grad, vox, gt = _ray_batch(R, S, 3)
    e = lambda *s: torch.full(s, float("nan"), device="cuda")
    # the three plain launches
    w, rgbv, dep, nm, ent = e(R, S), e(R, 3), e(R), e(R, 3), e(R)
    check(lib.nsa_composite_forward(rays_o.data_ptr(), rays_d.data_ptr(), z.data_ptr(), sdf.data_ptr(), rgb.data_ptr(),
                                    grad.data_ptr(), vox.data_ptr(), 32, R, S, w.data_ptr(), rgbv.data_ptr(), dep.data_ptr(),
                                    nm.data_ptr(), ent.data_ptr(), _st()))
    loss, g_rgbv = e(1), e(R, 3)
    check(lib.nsa_l1_loss(rgbv.data_ptr(), gt.data_ptr(), 3 * R, loss.data_ptr(), g_rgbv.data_ptr(), _st()))
    g_sdf, g_rgb, g_grad = e(R, S), e(R, S, 3), e(R, S, 3)
    check(lib.nsa_composite_backward(rays_o.data_ptr(), rays_d.data_ptr(), z.data_ptr(), sdf.data_ptr(), rgb.data_ptr(),
                                     grad.data_ptr(), vox.data_ptr(), 32, R, S, g_rgbv.data_ptr(), None, None, None, None,
                                     g_sdf.data_ptr(), g_rgb.data_ptr(), g_grad.data_ptr(), _st()))
    # one launch
    rgbv2, rl, g_sdf2, g_rgb2, g_grad2 = e(R, 3), e(R), e(R, S), e(R, S, 3), e(R, S, 3)
    check(lib.nsa_composite_track(rays_o.data_ptr(), rays_d.data_ptr(), z.data_ptr(), sdf.data_ptr(), rgb.data_ptr(),
                                  vox.data_ptr(), 32, R, S, gt.data_ptr(), R, rgbv2.data_ptr(), rl.data_ptr(), g_sdf2.data_ptr(),
                                  g_rgb2.data_ptr(), g_grad2.data_ptr(), _st()))
    torch.cuda.synchronize()
    assert torch.equal(rgbv2, rgbv)
    assert torch.equal(g_rgb2, g_rgb)
    assert torch.equal(g_sdf2, g_sdf) and bool(g_sdf.abs().max() > 0)
    assert torch.equal(g_grad2, torch.zeros_like(g_grad2)) and bool((g_grad == 0).all())
    assert torch.equal(rl, (rgbv - gt).abs()[:, 0] + (rgbv - gt).abs()[:, 1] + (rgbv - gt).abs()[:, 2])
    assert abs(float(rl.double().sum() / (3 * R)) - float(loss)) < 1e-6


@pytest.mark.parametrize("R,S,weight", [(1024, 128, 0.0), (1024, 128, 1024.0), (37, 98, 0.0), (3, 64, 3.0)])
def test_track_finish_is_ray_reduction_plus_tail(R, S, weight):
    from nicer_slam_amd._native import lib, check
    g = torch.Generator(device="cuda").manual_seed(5)
    r = lambda *s: torch.rand(*s, device="cuda", generator=g)
    z = torch.sort(r(R, S) * 1.5 + 0.05, dim=1).values.contiguous()
    g_x, g_dir = (r(R, S, 3) - 0.5).contiguous(), (r(R, S, 3) - 0.5).contiguous() * 0.1
    uv = torch.stack([r(R) * 1200, r(R) * 680], -1).contiguous()
    K = torch.eye(4, device="cuda")
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    cam0 = torch.tensor([0.98, 0.05, -0.1, 0.02, 0.1, 0.0, -0.2], device="cuda")
    ray_loss = r(R).contiguous()
    hyper = (0.005, 0.9, 0.999, 1e-8, 3, 0.5)
    do_adam = weight == 0.0

    def state():
        return dict(cam=cam0.clone(), m=torch.zeros(7, device="cuda"), v=torch.zeros(7, device="cuda"),
                    t=torch.zeros(1, device="cuda"), red=torch.zeros(9, device="cuda"), best=torch.tensor([1e10] + [0.0] * 7, device="cuda"))

    # plain: ray sums, loss slot, tail
    a = state()
    g_o, g_d = torch.empty(R, 3, device="cuda"), torch.empty(R, 3, device="cuda")
    for _ in range(2):
        check(lib.nsa_rays_backward(z.data_ptr(), g_x.data_ptr(), g_dir.data_ptr(), R, S, g_o.data_ptr(), g_d.data_ptr(), _st()))
        a["red"][7] = ray_loss.sum() / (3 * R)
        check(lib.nsa_track_tail(uv.data_ptr(), K.data_ptr(), a["cam"].data_ptr(), R, g_o.data_ptr(), g_d.data_ptr(),
                                 a["red"].data_ptr(), 1 if do_adam else 0, weight, a["m"].data_ptr(), a["v"].data_ptr(),
                                 a["t"].data_ptr(), *hyper, a["red"][7:8].data_ptr() if do_adam else None,
                                 a["best"].data_ptr() if do_adam else None, _st()))
    # folded
    outs = []
    for rep in range(3):
        b = state()
        ws = torch.zeros(int(lib.nsa_track_finish_workspace(R)), device="cuda")
        for _ in range(2):
            check(lib.nsa_track_finish(uv.data_ptr(), K.data_ptr(), b["cam"].data_ptr(), R, S, z.data_ptr(), g_x.data_ptr(),
                                       g_dir.data_ptr(), ray_loss.data_ptr(), b["red"].data_ptr(), 1 if do_adam else 0, weight,
                                       b["m"].data_ptr(), b["v"].data_ptr(), b["t"].data_ptr(), *hyper,
                                       b["best"].data_ptr() if do_adam else None, ws.data_ptr(), _st()))
        torch.cuda.synchronize()
        assert int(ws[:1].view(torch.int32)) == 0, "the ticket must be left at zero"
        outs.append(b)
    b = outs[0]
    scale = float(a["red"][:7].abs().max())
    assert_close(b["red"], a["red"], 2e-5 * scale, 2e-5, "message / gradient")
    assert_close(b["cam"], a["cam"], 1e-6, 1e-5, "camera after two steps")
    assert_close(b["best"], a["best"], 1e-6, 1e-5, "arg-min-loss candidate")
    assert float(b["t"]) == float(a["t"]) == (2.0 if do_adam else 0.0)
    for o in outs[1:]:                                   # fixed summation order: reproducible bit for bit
        for k in ("red", "cam", "m", "v", "best"):
            assert torch.equal(o[k], b[k]), k


def test_folded_tracker_follows_the_plain_sequence_and_is_reproducible():
    from nicer_slam_amd.tracking import KernelTracker
    fx, model, cam, pose, _, _ = _setup("full_tracking")
    model.train(True)
    model.engine = "fused"
    model.draws = draws_of(fx, "cuda")
    K, uv, gt = tt(fx["in_K"]).cuda(), tt(fx["in_uv"]).cuda(), tt(fx["gt_rgb"]).cuda()
    cam0 = tt(fx["in_cam"]).reshape(-1)
    runs = {}
    for tag, fold, use_graph in (("plain", "0", True), ("fold", "1", True), ("fold2", "1", True), ("fold_eager", "1", False)):
        kt = KernelTracker(model, K, uv.shape[1], cam0, lr=0.005, use_graph=use_graph, fold=fold == "1")
        assert kt.folded == (fold == "1")
        ls = [float(kt.step(uv, gt)) for _ in range(6)]
        runs[tag] = (torch.tensor(ls), kt.cam.clone(), kt.candidate.clone())
    for tag in ("fold", "fold_eager"):
        assert_close(runs[tag][0], runs["plain"][0], 1e-6, 1e-5, f"losses ({tag})")
        assert_close(runs[tag][1], runs["plain"][1], 1e-6, 1e-5, f"camera after 6 steps ({tag})")
        assert_close(runs[tag][2], runs["plain"][2], 1e-6, 1e-5, f"candidate ({tag})")
    assert torch.equal(runs["fold"][0], runs["fold2"][0]) and torch.equal(runs["fold"][1], runs["fold2"][1])
    assert torch.equal(runs["fold"][1], runs["fold_eager"][1])


def test_two_thousand_replays_stay_finite_reproducible_and_leave_the_tickets_at_zero():
    """Soak: the last-workgroup tickets of nsa_track_finish and nsa_draw are re-armed by the kernels themselves; 2000 graph replays of
    two identically seeded trackers must agree bit for bit, stay finite, and leave both tickets at zero and the call counter at the
    number of sampler calls."""
    from nicer_slam_amd.tracking import KernelTracker
    from nicer_slam_amd.fused import sampler as fs
    fx, model, cam, pose, _, _ = _setup("full_tracking")
    model.train(True)
    model.engine = "fused"
    model.draws = None
    K, uv, gt = tt(fx["in_K"]).cuda(), tt(fx["in_uv"]).cuda(), tt(fx["gt_rgb"]).cuda()
    cam0 = tt(fx["in_cam"]).reshape(-1)
    outs = []
    for rep in range(2):
        for k in ("_draw_state", "_draw_states", "_draw_seed"):
            model.__dict__.pop(k, None)
        torch.manual_seed(5)
        kt = KernelTracker(model, K, uv.shape[1], cam0, lr=0.0005, use_graph=True)
        calls0 = int(fs.draw_state(model)[1])
        losses = torch.stack([kt.step(uv, gt).clone() for _ in range(2000)])
        torch.cuda.synchronize()
        assert bool(torch.isfinite(losses).all()) and bool(torch.isfinite(kt.cam).all())
        st = fs.draw_state(model).cpu().tolist()
        assert st[1] == calls0 + 2000 and st[2] == 0
        assert int(kt.fin_ws[:1].view(torch.int32)) == 0
        assert float(kt.t) == 2000.0 + float(0)                       # Adam steps of this tracker since construction (reset by capture)
        outs.append((losses, kt.cam.clone(), kt.candidate.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("chunks,use_graph", [(2, True), (3, True), (3, False)])
def test_chunked_tracker_draws_from_one_state_per_chunk(chunks, use_graph):
    """KernelTracker(chunks > 1) with UNPINNED draws: the chunks run nsa_draw concurrently on forked streams, so each
    owns a {seed, call, ticket} state (fused/sampler.py::draw_state, keyed by the chunk's first row).  After N iterations every
    state must show N more calls and a ticket back at zero (a shared state ends with a non-zero ticket or a wrong count as soon as
    two launches overlap), the chunks' seeds must differ (no chunk repeats another's jitter), and the run must be reproducible."""
    from nicer_slam_amd.fused import sampler as fs
    from nicer_slam_amd.tracking import KernelTracker
    fx, model, cam, pose, _, _ = _setup("full_tracking")
    model.train(True)
    model.engine = "fused"
    model.draws = None
    K, uv, gt = tt(fx["in_K"]).cuda(), tt(fx["in_uv"]).cuda(), tt(fx["gt_rgb"]).cuda()
    cam0 = tt(fx["in_cam"]).reshape(-1)
    outs = []
    for rep in range(2):
        for k in ("_draw_state", "_draw_states", "_draw_seed"):
            model.__dict__.pop(k, None)
        torch.manual_seed(11)
        kt = KernelTracker(model, K, uv.shape[1], cam0, lr=0.0005, use_graph=use_graph, chunks=chunks)
        keys = [lo for lo, _ in kt.bounds]
        states = model.__dict__["_draw_states"] if use_graph else None
        if states is None:                        # eager: the states appear with the first iteration
            kt.step(uv, gt)
            states = model.__dict__["_draw_states"]
        assert sorted(states) == sorted(keys), (sorted(states), keys)
        before = {k: int(states[k][1]) for k in keys}
        N = 300
        losses = torch.stack([kt.step(uv, gt).clone() for _ in range(N)])
        torch.cuda.synchronize()
        assert bool(torch.isfinite(losses).all())
        seeds = set()
        for k in keys:
            st = states[k].cpu().tolist()
            assert st[1] == before[k] + N and st[2] == 0, (k, st, before[k])
            seeds.add(st[0])
        assert len(seeds) == chunks
        outs.append((losses, kt.cam.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("use_graph", [True, False])
def test_head_launch_draws_change_no_bit(use_graph):
    """The sampler's draws made by the tracker's head launch (nsa_track_begin_draw) instead of an nsa_draw node of their own, at the
    shipped sample counts (640 sampler evaluations, 128 samples per ray): the same generator state, the same numbers -- losses, camera,
    Adam moments, candidate and message of 8 iterations must be IDENTICAL, and the generator's call counter advances once per step."""
    from nicer_slam_amd.fused import sampler as fs
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.tracking import KernelTracker
    from nicer_slam_amd.utils.conf import replica_model_conf
    torch.manual_seed(0)
    model = SLAMNetwork(replica_model_conf(94, 640, 32, use_warp_loss=False), n_images=1,
                        colour_grid=dict(base_resolution=16, desired_resolution=256, log2_hashmap_size=14)).cuda().train()
    for p in model.parameters():
        p.requires_grad_(False)
    g = torch.Generator(device="cuda").manual_seed(3)
    with torch.no_grad():
        for enc in (model.implicit_network.coarse.encoding, model.implicit_network.fine.encoding, model.rendering_network.encoding):
            enc.embeddings.copy_((torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1) * 0.05)
    R = 256
    K = torch.eye(4, device="cuda")
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    batches = [(torch.stack([torch.rand(R, device="cuda", generator=g) * 1199, torch.rand(R, device="cuda", generator=g) * 679], -1)[None],
                torch.rand(R, 3, device="cuda", generator=g)) for _ in range(8)]
    cam0 = torch.tensor([1.0, 0.02, -0.01, 0.03, 0.1, 0.0, -0.2], device="cuda")
    runs = {}
    for tag, in_begin in (("node", False), ("head", True)):
        for k in ("_draw_state", "_draw_states", "_draw_seed"):
            model.__dict__.pop(k, None)
        torch.manual_seed(7)
        kt = KernelTracker(model, K[None], R, cam0, lr=0.002, use_graph=use_graph, draw_in_begin=in_begin)
        assert (kt.drawn is not None) == (tag == "head")
        calls0 = int(fs.draw_state(model)[1])
        ls = torch.stack([kt.step(*b).clone() for b in batches])
        torch.cuda.synchronize()
        st = fs.draw_state(model).cpu().tolist()
        assert st[1] == calls0 + len(batches) and st[2] == 0
        runs[tag] = (ls, kt.cam.clone(), kt.m.clone(), kt.v.clone(), kt.best.clone(), kt.red.clone())
    assert bool(torch.isfinite(runs["node"][0]).all()) and not torch.equal(runs["node"][1], cam0)
    for a, b, what in zip(runs["head"], runs["node"], ("losses", "camera", "exp_avg", "exp_avg_sq", "candidate", "message")):
        assert torch.equal(a, b), f"{what} differ by {float((a - b).abs().max()):.3g}"


@pytest.mark.parametrize("use_graph", [True, False])
def test_colour_forward_with_the_ray_objective_in_its_launch_changes_no_bit(use_graph):
    """nsa_colour_forward_track (at 128 samples per ray a workgroup of the colour forward is one ray: its first wave runs the composite +
    L1 + composite backward of that ray when the colours are stored) against nsa_colour_forward + nsa_composite_track: losses, camera,
    Adam moments, candidate and message of 8 tracker iterations must be IDENTICAL."""
    from nicer_slam_amd.fused import render as fr
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.tracking import KernelTracker
    from nicer_slam_amd.utils.conf import replica_model_conf
    torch.manual_seed(0)
    model = SLAMNetwork(replica_model_conf(94, 640, 32, use_warp_loss=False), n_images=1,
                        colour_grid=dict(base_resolution=16, desired_resolution=256, log2_hashmap_size=14)).cuda().train()
    for p in model.parameters():
        p.requires_grad_(False)
    g = torch.Generator(device="cuda").manual_seed(3)
    with torch.no_grad():
        for enc in (model.implicit_network.coarse.encoding, model.implicit_network.fine.encoding, model.rendering_network.encoding):
            enc.embeddings.copy_((torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1) * 0.05)
    R = 256
    K = torch.eye(4, device="cuda")
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    batches = [(torch.stack([torch.rand(R, device="cuda", generator=g) * 1199, torch.rand(R, device="cuda", generator=g) * 679], -1)[None],
                torch.rand(R, 3, device="cuda", generator=g)) for _ in range(8)]
    cam0 = torch.tensor([1.0, 0.02, -0.01, 0.03, 0.1, 0.0, -0.2], device="cuda")
    runs = {}
    assert fr.COLOUR_FWD_TRACK
    for merged in (False, True):
        for k in ("_draw_state", "_draw_states", "_draw_seed"):
            model.__dict__.pop(k, None)
        torch.manual_seed(7)
        fr.COLOUR_FWD_TRACK = merged
        try:
            kt = KernelTracker(model, K[None], R, cam0, lr=0.002, use_graph=use_graph)
            ls = torch.stack([kt.step(*b).clone() for b in batches])
            torch.cuda.synchronize()
        finally:
            fr.COLOUR_FWD_TRACK = True
        runs[merged] = (ls, kt.cam.clone(), kt.m.clone(), kt.v.clone(), kt.best.clone(), kt.red.clone())
    assert bool(torch.isfinite(runs[False][0]).all()) and not torch.equal(runs[False][1], cam0)
    for a, b, what in zip(runs[True], runs[False], ("losses", "camera", "exp_avg", "exp_avg_sq", "candidate", "message")):
        assert torch.equal(a, b), f"{what} differ by {float((a - b).abs().max()):.3g}"
