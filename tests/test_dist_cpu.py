"""N > 1 path on CPU: world_size-2 gloo, ray-sharded tracking step, fused 9-float all-reduce of the pose gradient.
The sharded result must equal the single-process gradient captured from the reference (goldens)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import load, tt, draws_of


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import hashenc
    from nicer_slam_amd.hashencoder import hashgrid
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    from nicer_slam_amd import dist as nd
    from test_model_cpu import build_model
    hashgrid._backend = hashenc.OracleBackend()       # CPU stand-in at the native seam (tests only)
    fx = load("full_tracking")
    model = build_model(fx)
    model.train(True)
    n = fx["in_uv"].shape[1]
    lo, hi = nd.shard_rays(n, rank, world)
    d = draws_of(fx)
    model.draws = {"t_rand": d["t_rand"][lo:hi], "extra_idx": d["extra_idx"], "eik_idx": d["eik_idx"][lo:hi],
                   "z_vals_override": tt(fx["out_z_vals"])[lo:hi]}
    cam = tt(fx["in_cam"]).requires_grad_(True)
    out = model({"intrinsics": tt(fx["in_K"]), "uv": tt(fx["in_uv"])[:, lo:hi], "pose": get_camera_from_tensor(cam)},
                torch.arange(1), {}, mode="tracking", frame_idx=1)
    loss = (out["rgb_values"].reshape(-1, 3) - tt(fx["gt_rgb"])[lo:hi]).abs().mean()
    loss.backward()
    g, l = nd.allreduce_pose_grad(cam.grad.reshape(-1), loss, hi - lo)
    out_q.put((rank, g.numpy().copy(), float(l)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_rays_balanced():
    from nicer_slam_amd.dist import shard_rays
    for n, w in [(1024, 8), (1000, 3), (7, 2), (5, 8)]:
        spans = [shard_rays(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_sharded_tracking_step_matches_reference_gradient():
    fx = load("full_tracking")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=0, atol=0)          # identical on every rank
    ref = fx["grad_cam"].reshape(-1)
    np.testing.assert_allclose(res[0][1], ref, rtol=1e-3, atol=1e-6 + 1e-4 * np.abs(ref).max())
    assert abs(res[0][2] - float(fx["out_loss"])) < 1e-5


# ---------------------------------------------------------------------------------------------- mapping exchange
def _torch_math_adam(p, g, m, v, step, lr, betas, eps):
    """Test-only stepper (torch.optim.Adam's update written out) so ShardedAdam's collectives can run on CPU tensors;
    the product default is the HIP kernel."""
    b1, b2 = betas
    if g is None:                # none_grad="zeros": the zero-gradient step (what torch 1.11 does after zero_grad())
        g = torch.zeros_like(p)
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = (v.sqrt() / (1 - b2 ** step) ** 0.5).add_(eps)
    p.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))


def _rank_grads(rank, it, shapes):
    g = torch.Generator().manual_seed(1000 * it + rank)
    return [torch.randn(*s, generator=g) * (0.1 + it) for s in shapes]


_SHAPES = [(37,), (70001, 3), (5, 7), (40000, 2)]   # small, sharded (odd size -> padded staging buffer), small, sharded (even:
                                                   # collectives run straight on the gradient / parameter storage)
_WEIGHTS = [0.25, 0.75]                          # unequal ray shares
_ITERS = 5
# (iteration, parameter) pairs without a gradient -- a table outside its stage (volsdf_train.py:550-555): one sharded + padded tensor, one
# small one.  With none_grad="zeros" they take torch 1.11's zero-gradient step; the single-process reference is fed explicit zeros.
_NO_GRAD = {(3, 1), (3, 2)}


def _map_worker(rank, world, port, out_q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from nicer_slam_amd import dist as nd
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(*s)) for s in _SHAPES]
    opt = nd.ShardedAdam([{"params": params[:2], "lr": 0.04}, {"params": params[2:], "lr": 0.002}], betas=(0.9, 0.99),
                         eps=1e-15, stepper=_torch_math_adam, shard_min_numel=1 << 16, none_grad="zeros")
    ptrs = None
    for it in range(_ITERS):
        for i, (p, g) in enumerate(zip(params, _rank_grads(rank, it, _SHAPES))):
            p.grad = None if (it, i) in _NO_GRAD else g          # (what zero_grad() leaves under torch >= 2.0)
        opt.step(weight=_WEIGHTS[rank])
        # no buffer churn: every persistent buffer of the sharded tensors keeps its storage from the first step on
        now = [opt.state[params[i]][k].data_ptr() for i in (1, 3) for k in ("exp_avg", "exp_avg_sq", "g_shard")]
        now.append(opt.state[params[1]]["flat"].data_ptr())
        assert ptrs is None or now == ptrs
        ptrs = now
    assert opt.state[params[1]]["sharded"] and opt.state[params[1]]["exp_avg"].numel() == 105002 and opt.state[params[1]]["padded"]
    assert opt.state[params[3]]["sharded"] and not opt.state[params[3]]["padded"] and "flat" not in opt.state[params[3]]
    assert not opt.state[params[0]]["sharded"]
    vox0 = torch.arange(8.0).reshape(2, 2, 2)
    vox = vox0 + (rank + 1) * torch.tensor([1.0, 0, 0, 2, 0, 0, 0, 3]).reshape(2, 2, 2)
    nd.allreduce_voxel_delta(vox, vox0)
    out_q.put((rank, [p.detach().numpy().copy() for p in params], vox.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_adam_matches_single_process_adam():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_map_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    ref = [torch.nn.Parameter(torch.randn(*s)) for s in _SHAPES]
    opt = torch.optim.Adam([{"params": ref[:2], "lr": 0.04}, {"params": ref[2:], "lr": 0.002}], betas=(0.9, 0.99), eps=1e-15)
    for it in range(_ITERS):
        gs = [_rank_grads(r, it, _SHAPES) for r in range(2)]
        for i, p in enumerate(ref):
            p.grad = torch.zeros_like(p) if (it, i) in _NO_GRAD else _WEIGHTS[0] * gs[0][i] + _WEIGHTS[1] * gs[1][i]
        opt.step()
    for a, b, r in zip(res[0][1], res[1][1], ref):
        np.testing.assert_array_equal(a, b)                                   # replicas stay bit-identical
        np.testing.assert_allclose(a, r.detach().numpy(), rtol=2e-5, atol=2e-6)
    expect = np.arange(8.0).reshape(2, 2, 2) + 3 * np.array([1.0, 0, 0, 2, 0, 0, 0, 3]).reshape(2, 2, 2)
    np.testing.assert_array_equal(res[0][2], expect)
    np.testing.assert_array_equal(res[1][2], expect)


# ---------------------------------------------------------------------------------------------- SSI depth sums
def _ssi_inputs():
    g = torch.Generator().manual_seed(5)
    b, n = 3, 40
    pred = torch.rand(b, n, 1, generator=g) * 2 + 0.1
    target = 1.7 * pred + 0.3 + 0.05 * torch.randn(b, n, 1, generator=g)
    mask = torch.rand(b, n, 1, generator=g) > 0.3
    return pred, target, mask


def _ssi_worker(rank, world, port, out_q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from nicer_slam_amd.dist import shard_rays
    from nicer_slam_amd.model.loss import scale_shift_invariant_depth_loss
    pred, target, mask = _ssi_inputs()
    lo, hi = shard_rays(pred.shape[1], rank, world)          # rays sharded WITHIN every keyframe (unequal: 40 = 14 + 13 + 13)
    w = (hi - lo) / pred.shape[1]
    p = pred[:, lo:hi].clone().requires_grad_(True)
    loss = scale_shift_invariant_depth_loss(p, target[:, lo:hi], mask[:, lo:hi], alpha=0.0, shard=(None, w))
    loss.backward()
    out_q.put((rank, lo, hi, w, float(loss), p.grad.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ray_sharded_ssi_depth_loss_equals_single_process():
    """The five per-image sums of the scale-and-shift solve (MiDaS.py:6-26) all-reduced over ranks: the weighted sum of the
    ranks' losses and gradients is the single-process loss and gradient (data term; alpha = 0 -- the first-difference
    regulariser drops the pairs that straddle a shard boundary by construction)."""
    from nicer_slam_amd.model.loss import scale_shift_invariant_depth_loss
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ssi_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pred, target, mask = _ssi_inputs()
    pr = pred.clone().requires_grad_(True)
    ref = scale_shift_invariant_depth_loss(pr, target, mask, alpha=0.0)
    ref.backward()
    total = sum(w * l for _, _, _, w, l, _ in res)
    assert abs(total - float(ref)) < 1e-6 * max(1.0, abs(float(ref)))
    grad = torch.cat([torch.from_numpy(g) * w for _, _, _, w, _, g in res], dim=1)
    # (the 2x2 solve sees the five sums in another summation order: ~1e-7 relative in scale / shift)
    np.testing.assert_allclose(grad.numpy(), pr.grad.numpy(), rtol=1e-4, atol=1e-4 * float(pr.grad.abs().max()))
