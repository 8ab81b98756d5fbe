"""Multi-GPU execution of the render core (new -- the reference is single-process).

Rays are independent given replicated parameters, so the path shards by rays with no data-path collective: one
process per GPU (RCCL over xGMI = torch.distributed backend "nccl"), each rank renders its own contiguous ray shard.
A tracking step exchanges exactly one message: the 7-float pose gradient + the loss, averaged over ranks (the
objective is a mean over the global ray batch, code/model/loss.py:57-65)."""
import torch
import torch.distributed as dist

def _bump_version(p):
    """The kernel wrote p behind autograd's back: advance its version counter like an in-place op would (the packed-weight
    caches of the fused engine key on it).  Private torch API with a portable fallback for the small tensors that matter."""
    try:
        torch._C._autograd._unsafe_set_version_counter((p,), (p._version + 1,))
    except Exception:                      # older / newer torch: an in-place no-op does the same for MLP-sized tensors
        if p.numel() <= (1 << 20):
            p.add_(0)



def shard_rays(n_rays, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_rays for `rank`."""
    base, rem = divmod(n_rays, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_buf = {}


def allreduce_pose_grad(cam_grad, loss, n_local, group=None):
    """Average the pose gradient and the loss over the global ray batch with ONE fused all-reduce.

    cam_grad [7] and loss are this rank's values for the mean over its n_local rays; ranks may hold different ray
    counts, so each contribution is weighted by n_local / n_global."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return cam_grad, loss
    key = (cam_grad.device, cam_grad.dtype)
    buf = _buf.get(key)
    if buf is None:
        buf = _buf[key] = torch.zeros(9, device=cam_grad.device, dtype=cam_grad.dtype)
    buf[:7] = cam_grad * n_local
    buf[7] = loss.detach() * n_local
    buf[8] = float(n_local)
    dist.all_reduce(buf, group=group)
    return buf[:7] / buf[8], buf[7] / buf[8]


# ---------------------------------------------------------------------------------------------- mapping
def allreduce_voxel_delta(voxels, before, group=None):
    """Make the visit counter global: every rank counted only its own ray shard's samples into ``voxels`` since
    ``before``; sum the deltas.  (update_voxels happens before the density lookup of the same forward, network.py:
    62-76, so SLAMNetwork calls this between the two -- see SLAMNetwork.voxel_sync.)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return voxels
    delta = voxels - before
    dist.all_reduce(delta, group=group)
    voxels.copy_(before + delta)
    return voxels


def _hip_adam(p, g, exp_avg, exp_avg_sq, step, lr, betas, eps):
    from ._native import lib, check
    if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
        raise RuntimeError("ShardedAdam: the built-in stepper needs float32 contiguous CUDA tensors")
    check(lib.nsa_adam_table_step(p.data_ptr(), g.contiguous().data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                  p.numel(), int(step), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                  torch.cuda.current_stream().cuda_stream))


class ShardedAdam(torch.optim.Optimizer):
    """Data-parallel Adam for a mapping step: gradients of replicated parameters are averaged over ranks and the
    parameters stay bit-identical on every rank.

    * small tensors (MLPs; < ``shard_min_numel``): ONE bucketed all-reduce of all their gradients, full Adam everywhere.
    * large tensors (the grid tables; the colour table is 1 GiB): reduce-scatter the gradient, run Adam on this rank's
      1/world slice only (moments exist only for the slice: 1/world of the optimizer memory and HBM traffic), then
      all-gather the updated slices.  On xGMI (7 point-to-point links per GPU) both collectives move (world-1)/world of
      the tensor per rank, spread over all links -- the exchange SURVEY 8e recommends over a ring all-reduce.

    ``weight`` (this rank's share of the global ray batch, e.g. n_local / n_global) scales the local gradient before
    the SUM so that ranks with different ray counts still produce the gradient of the global mean.
    Hyper-parameters/semantics as nicer_slam_amd.optim.Adam / torch.optim.Adam (no weight decay, no amsgrad).
    ``stepper`` = callable(p, g, exp_avg, exp_avg_sq, step, lr, betas, eps) updating p in place; default: the HIP kernel.
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, group=None, shard_min_numel=1 << 16, stepper=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.group, self.shard_min_numel, self.stepper = group, shard_min_numel, stepper or _hip_adam
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0

    def _reduce_scatter(self, flat, shard):
        if dist.get_backend(self.group) == "gloo":      # gloo has no reduce-scatter: same result, more bytes
            dist.all_reduce(flat, group=self.group)
            shard.copy_(flat.view(self.world, -1)[self.rank])
        else:
            dist.reduce_scatter_tensor(shard, flat, group=self.group)

    @torch.no_grad()
    def step(self, weight=None):
        w = (1.0 / self.world) if weight is None else float(weight)
        small = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                state = self.state[p]
                if not state:
                    sharded = self.world > 1 and p.numel() >= self.shard_min_numel
                    n = -(-p.numel() // self.world) if sharded else p.numel()
                    state.update(step=0, sharded=sharded, shard_numel=n,
                                 exp_avg=torch.zeros(n, device=p.device, dtype=p.dtype),
                                 exp_avg_sq=torch.zeros(n, device=p.device, dtype=p.dtype))
                state["step"] += 1
                if state["sharded"]:
                    n = state["shard_numel"]
                    flat = torch.zeros(n * self.world, device=p.device, dtype=p.dtype)
                    flat[:p.numel()] = p.grad.reshape(-1) * w
                    g_shard = torch.empty(n, device=p.device, dtype=p.dtype)
                    self._reduce_scatter(flat, g_shard)
                    flat[:p.numel()] = p.reshape(-1)                   # reuse as the parameter gather buffer
                    p_shard = flat.view(self.world, n)[self.rank].clone()
                    self.stepper(p_shard, g_shard, state["exp_avg"], state["exp_avg_sq"], state["step"], group["lr"],
                                 group["betas"], group["eps"])
                    dist.all_gather_into_tensor(flat, p_shard, group=self.group)
                    p.copy_(flat[:p.numel()].view_as(p))
                else:
                    small.append((p, group, state))
        if small:
            if self.world > 1 or w != 1.0:
                bucket = torch.cat([p.grad.reshape(-1) * w for p, _, _ in small])
                if self.world > 1:
                    dist.all_reduce(bucket, group=self.group)
                o = 0
                for p, _, _ in small:
                    p.grad.copy_(bucket[o:o + p.numel()].view_as(p))
                    o += p.numel()
            for p, group, state in small:
                flat_p = p.view(-1)
                self.stepper(flat_p, p.grad.reshape(-1), state["exp_avg"], state["exp_avg_sq"], state["step"],
                             group["lr"], group["betas"], group["eps"])
                _bump_version(p)
