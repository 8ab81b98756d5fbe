"""Multi-GPU execution of the render core (new -- the reference is single-process).

Rays are independent given replicated parameters, so the path shards by rays with no data-path collective: one
process per GPU (RCCL over xGMI = torch.distributed backend "nccl"), each rank renders its own contiguous ray shard.
A tracking step exchanges exactly one message: the 7-float pose gradient + the loss, averaged over ranks (the
objective is a mean over the global ray batch, code/model/loss.py:57-65)."""
import torch
import torch.distributed as dist

from ._version import bump_version



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
    if g is None:            # the zero-gradient step of none_grad="zeros" (optim.py): no gradient is read
        check(lib.nsa_adam_table_step_zero_grad(p.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), p.numel(), int(step), float(lr),
                                                float(betas[0]), float(betas[1]), float(eps), torch.cuda.current_stream().cuda_stream))
        return
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
    ``none_grad`` as nicer_slam_amd.optim.Adam: "skip" (installed torch) or "zeros" (torch 1.11, the reference's environment) -- a
    parameter whose ``.grad`` is None on this step but which has been stepped before takes a zero-gradient step (``stepper`` is called
    with ``g = None``).  Which parameters receive gradients is decided by the mapping schedule (stage / color_stage), i.e. identically on
    every rank, so no collective is needed for the gradient: each rank steps its slice, the slices are all-gathered as usual.

    ``p.grad`` is CONSUMED by ``step()``: it is scaled by ``weight`` in place and the collectives run on its storage (the
    reduce-scatter reads it; under gloo the fallback all-reduces into it), so after the step it no longer holds this rank's
    local gradient -- call ``zero_grad()`` before the next backward as usual, and read local gradients before ``step()``.
    Branches that have never run with more than one rank on real hardware (no multi-GPU box was available to any build round):
    the ``nccl`` in-place forms ``dist.reduce_scatter_tensor`` / ``dist.all_gather_into_tensor`` on views of the gradient /
    parameter storage (``_reduce_scatter`` / ``_all_gather``); the gloo tests (world 2 and 3) take the all-reduce / all-gather
    list fallbacks of the same methods, and a 1-rank RCCL group exercises the in-place forms with world = 1 only.
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, group=None, shard_min_numel=1 << 16, stepper=None,
                 none_grad="skip"):
        if none_grad not in ("skip", "zeros"):
            raise ValueError(f"none_grad={none_grad!r}: expected 'skip' or 'zeros'")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.none_grad = none_grad
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

    def _all_gather(self, flat, shard):
        """flat[world * n] <- every rank's shard[n]; ``shard`` may be the rank's own slice of ``flat`` (in place: RCCL/NCCL
        define that case; gloo gets a private copy of the input)."""
        if dist.get_backend(self.group) == "gloo":
            shard = shard.clone()
        dist.all_gather_into_tensor(flat, shard, group=self.group)

    def _sharded_state(self, p, state):
        """Persistent buffers of one sharded tensor (allocated once, on the first step): the rank's moment slices and gradient
        slice and -- only when numel is not a multiple of world -- one padded staging buffer whose tail stays zero."""
        n = -(-p.numel() // self.world)
        state.update(step=0, sharded=True, shard_numel=n, padded=n * self.world != p.numel(),
                     exp_avg=torch.zeros(n, device=p.device, dtype=p.dtype),
                     exp_avg_sq=torch.zeros(n, device=p.device, dtype=p.dtype),
                     g_shard=torch.empty(n, device=p.device, dtype=p.dtype))
        if state["padded"]:
            state["flat"] = torch.zeros(n * self.world, device=p.device, dtype=p.dtype)

    @torch.no_grad()
    def step(self, weight=None):
        w = (1.0 / self.world) if weight is None else float(weight)
        small = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    state = self.state.get(p)
                    if getattr(self, "none_grad", "skip") != "zeros" or not state:
                        continue
                    state["step"] += 1
                    if state["sharded"]:
                        n, numel = state["shard_numel"], p.numel()
                        if state["padded"]:
                            flat = state["flat"]
                            flat[:numel].copy_(p.view(-1))
                        else:
                            flat = p.view(-1)
                        p_shard = flat[self.rank * n:(self.rank + 1) * n]
                        self.stepper(p_shard, None, state["exp_avg"], state["exp_avg_sq"], state["step"], group["lr"], group["betas"],
                                     group["eps"])
                        self._all_gather(flat, p_shard)
                        if state["padded"]:
                            p.view(-1).copy_(flat[:numel])
                    else:
                        self.stepper(p.view(-1), None, state["exp_avg"], state["exp_avg_sq"], state["step"], group["lr"], group["betas"],
                                     group["eps"])
                    bump_version(p)
                    continue
                state = self.state[p]
                if not state:
                    if self.world > 1 and p.numel() >= self.shard_min_numel:
                        self._sharded_state(p, state)
                    else:
                        state.update(step=0, sharded=False, shard_numel=p.numel(),
                                     exp_avg=torch.zeros(p.numel(), device=p.device, dtype=p.dtype),
                                     exp_avg_sq=torch.zeros(p.numel(), device=p.device, dtype=p.dtype))
                state["step"] += 1
                if state["sharded"]:
                    # reduce-scatter the (weighted) gradient -> Adam on this rank's slice of the PARAMETER, in place ->
                    # all-gather the slices back into the parameter.  No per-step allocation: the gradient is scaled in place
                    # (it is consumed here) and, when numel is a multiple of world, every collective runs straight on the
                    # gradient / parameter storage; otherwise through the one persistent padded buffer.
                    n, numel = state["shard_numel"], p.numel()
                    if not (p.is_contiguous() and p.grad.is_contiguous()):
                        raise RuntimeError("ShardedAdam: sharded tensors and their gradients must be contiguous")
                    g = p.grad.view(-1).mul_(w)
                    if state["padded"]:
                        flat = state["flat"]
                        flat[:numel].copy_(g)
                        self._reduce_scatter(flat, state["g_shard"])
                        flat[:numel].copy_(p.view(-1))
                    else:
                        self._reduce_scatter(g, state["g_shard"])
                        flat = p.view(-1)
                    p_shard = flat[self.rank * n:(self.rank + 1) * n]
                    self.stepper(p_shard, state["g_shard"], state["exp_avg"], state["exp_avg_sq"], state["step"], group["lr"],
                                 group["betas"], group["eps"])
                    self._all_gather(flat, p_shard)
                    if state["padded"]:
                        p.view(-1).copy_(flat[:numel])
                    bump_version(p)
                else:
                    small.append((p, group, state))
        if small:
            if self.world > 1 or w != 1.0:
                n_small = sum(p.numel() for p, _, _ in small)
                bucket = self.__dict__.get("_bucket")
                if bucket is None or bucket.numel() != n_small or bucket.device != small[0][0].device:
                    bucket = self._bucket = torch.empty(n_small, device=small[0][0].device, dtype=small[0][0].dtype)
                o = 0
                for p, _, _ in small:
                    torch.mul(p.grad.reshape(-1), w, out=bucket[o:o + p.numel()])
                    o += p.numel()
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
                bump_version(p)
