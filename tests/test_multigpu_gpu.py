"""The N > 1 paths over RCCL on REAL GPUs: these tests arm themselves on any box with >= 2 devices and are skipped (not absent) on a
one-GPU box -- no build round has had a multi-GPU box, so the first one that does runs them.

Covered with 2 ranks, one process per GPU, backend "nccl" (= RCCL):
  * KernelTracker(world=2): ray-sharded tracking steps (eager, hipGraph, hipGraph with the 9-float all-reduce + Adam captured)
    against the single-GPU trajectory on the concatenated batch;
  * dist.ShardedAdam: reduce_scatter_tensor / all_gather_into_tensor IN PLACE on the gradient / parameter storage (even sizes) and
    through the padded staging buffer (odd sizes), the HIP Adam kernel on the rank's slice -- against single-process torch.optim.Adam;
  * dist.allreduce_voxel_delta, dist.allreduce_pose_grad.
The same maths runs on CPU under gloo in tests/test_dist_cpu.py (there the collectives take gloo's list fallbacks).
Reference: the reference is single-process; semantics = volsdf_train.py:406-446 / :150-174 on the global ray batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from helpers import load, tt, draws_of, assert_close

pytestmark = pytest.mark.gpu
needs_two = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL over xGMI); self-arming")
# "gloo": the same two-process test with both ranks sharing device 0 -- runs on the one-GPU boxes, so the test's own logic (and the
# world-2 KernelTracker path with two real processes) is exercised before a multi-GPU box ever sees it; no captured collective there
BACKENDS = [pytest.param("nccl", marks=needs_two), "gloo"]

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port, backend="nccl"):
    import sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist, dev


def _spawn(fn, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=fn, args=(r, WORLD, port, q) + args) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(WORLD)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


# ------------------------------------------------------------------------------------------------ tracking
VARIANTS = [("eager", False, False), ("graph", True, False), ("graph+captured collective", True, True)]


def _tracker_worker(rank, world, port, out_q, backend):
    dist, dev = _init(rank, world, port, backend)
    from nicer_slam_amd import dist as nd
    from nicer_slam_amd.tracking import KernelTracker
    from test_model_cpu import build_model
    fx = load("full_tracking")
    model = build_model(fx).to(dev)
    model.train(True)
    model.engine = "fused"
    model.voxels = tt(fx["in_voxels"]).to(dev)
    for p in model.parameters():
        p.requires_grad_(False)
    n = fx["in_uv"].shape[1]
    lo, hi = nd.shard_rays(n, rank, world)
    d = draws_of(fx, dev)
    model.draws = {"t_rand": d["t_rand"][lo:hi].contiguous(), "extra_idx": d["extra_idx"], "eik_idx": d["eik_idx"][lo:hi].contiguous()}
    K, uv, gt = tt(fx["in_K"]).to(dev), tt(fx["in_uv"]).to(dev)[:, lo:hi].contiguous(), tt(fx["gt_rgb"]).to(dev)[lo:hi].contiguous()
    cam0 = tt(fx["in_cam"]).reshape(-1)
    out = {}
    for name, graph, captured in VARIANTS:
        if captured and backend != "nccl":
            continue
        kt = KernelTracker(model, K, hi - lo, cam0, lr=0.005, use_graph=graph, world=world, graph_collective=captured)
        assert kt.collective_in_graph == captured
        losses = [float(kt.step(uv, gt)) for _ in range(4)]
        torch.cuda.synchronize()
        out[name] = (losses, kt.cam.cpu().numpy().copy(), kt.candidate.cpu().numpy().copy(), float(kt.red[8]))
    out_q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("backend", BACKENDS)
def test_two_rank_kernel_tracker_follows_the_single_gpu_trajectory(backend):
    from nicer_slam_amd.tracking import KernelTracker
    from test_fused_gpu import _setup
    res = _spawn(_tracker_worker, backend)
    fx, model, cam, pose, _, _ = _setup("full_tracking")
    model.train(True)
    model.engine = "fused"
    model.draws = draws_of(fx, "cuda")
    K, uv, gt = tt(fx["in_K"]).cuda(), tt(fx["in_uv"]).cuda(), tt(fx["gt_rgb"]).cuda()
    cam0 = tt(fx["in_cam"]).reshape(-1)
    one = KernelTracker(model, K, uv.shape[1], cam0, lr=0.005, use_graph=True)
    ref_losses = [float(one.step(uv, gt)) for _ in range(4)]
    for name, _, captured in VARIANTS:
        if captured and backend != "nccl":
            continue
        l0, c0, b0, n0 = res[0][1][name]
        l1, c1, b1, n1 = res[1][1][name]
        assert l0 == l1 and np.array_equal(c0, c1) and np.array_equal(b0, b1), f"{name}: the ranks disagree"   # replicas stay identical
        assert n0 == n1 == float(uv.shape[1]), f"{name}: ray count in the message"
        assert_close(torch.tensor(l0), torch.tensor(ref_losses), 1e-6, 1e-5, f"{name}: losses")
        assert_close(torch.from_numpy(c0), one.cam, 1e-6, 1e-5, f"{name}: camera after 4 steps")
        assert_close(torch.from_numpy(b0), one.candidate, 1e-6, 1e-5, f"{name}: arg-min-loss camera")


# ------------------------------------------------------------------------------------------------ mapping exchange
_SHAPES = [(37,), (70001, 3), (5, 7), (40000, 2), (1 << 20, 2)]   # small | sharded odd (padded staging) | small | sharded even (in place) x2
_WEIGHTS = [0.25, 0.75]


def _rank_grads(rank, it):
    g = torch.Generator().manual_seed(1000 * it + rank)
    return [torch.randn(*s, generator=g) * (0.1 + it) for s in _SHAPES]


def _adam_worker(rank, world, port, out_q):
    dist, dev = _init(rank, world, port)
    from nicer_slam_amd import dist as nd
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(*s).to(dev)) for s in _SHAPES]
    opt = nd.ShardedAdam([{"params": params[:2], "lr": 0.04}, {"params": params[2:], "lr": 0.002}], betas=(0.9, 0.99), eps=1e-15,
                         shard_min_numel=1 << 16)                      # default stepper: the HIP Adam kernel on the rank's slice
    ptrs = None
    for it in range(3):
        for p, g in zip(params, _rank_grads(rank, it)):
            p.grad = g.to(dev)
        opt.step(weight=_WEIGHTS[rank])
        now = [opt.state[params[i]][k].data_ptr() for i in (1, 3, 4) for k in ("exp_avg", "exp_avg_sq", "g_shard")]
        assert ptrs is None or now == ptrs, "per-step allocation in ShardedAdam"
        ptrs = now
    assert opt.state[params[1]]["sharded"] and opt.state[params[1]]["padded"]
    assert opt.state[params[3]]["sharded"] and not opt.state[params[3]]["padded"] and "flat" not in opt.state[params[3]]
    vox0 = torch.arange(8.0, device=dev).reshape(2, 2, 2)
    vox = vox0 + (rank + 1) * torch.tensor([1.0, 0, 0, 2, 0, 0, 0, 3], device=dev).reshape(2, 2, 2)
    nd.allreduce_voxel_delta(vox, vox0)
    g_cam = torch.arange(7.0, device=dev) * (rank + 1)
    g, l = nd.allreduce_pose_grad(g_cam, torch.tensor(float(rank + 1), device=dev), 100 * (rank + 1))
    torch.cuda.synchronize()
    out_q.put((rank, [p.detach().cpu().numpy().copy() for p in params], vox.cpu().numpy().copy(), g.cpu().numpy().copy(), float(l)))
    dist.barrier()
    dist.destroy_process_group()


@needs_two
@pytest.mark.timeout(900)
def test_rccl_two_rank_sharded_adam_and_small_exchanges():
    res = _spawn(_adam_worker)
    torch.manual_seed(0)
    ref = [torch.nn.Parameter(torch.randn(*s)) for s in _SHAPES]
    opt = torch.optim.Adam([{"params": ref[:2], "lr": 0.04}, {"params": ref[2:], "lr": 0.002}], betas=(0.9, 0.99), eps=1e-15)
    for it in range(3):
        gs = [_rank_grads(r, it) for r in range(WORLD)]
        for i, p in enumerate(ref):
            p.grad = _WEIGHTS[0] * gs[0][i] + _WEIGHTS[1] * gs[1][i]
        opt.step()
    for a, b, r in zip(res[0][1], res[1][1], ref):
        np.testing.assert_array_equal(a, b)                                   # replicas stay bit-identical
        np.testing.assert_allclose(a, r.detach().numpy(), rtol=2e-5, atol=2e-6)
    expect = np.arange(8.0).reshape(2, 2, 2) + 3 * np.array([1.0, 0, 0, 2, 0, 0, 0, 3]).reshape(2, 2, 2)
    for r in res:
        np.testing.assert_array_equal(r[2], expect)
        # weighted mean over the global batch: (100 * 1 * g + 200 * 2 * g) / 300, loss (100 * 1 + 200 * 2) / 300
        np.testing.assert_allclose(r[3], np.arange(7.0) * 5.0 / 3.0, rtol=1e-6)
        assert abs(r[4] - 5.0 / 3.0) < 1e-6
