"""Multi-GPU execution of the render core (new -- the reference is single-process).

Rays are independent given replicated parameters, so the path shards by rays with no data-path collective: one
process per GPU (RCCL over xGMI = torch.distributed backend "nccl"), each rank renders its own contiguous ray shard.
A tracking step exchanges exactly one message: the 7-float pose gradient + the loss, averaged over ranks (the
objective is a mean over the global ray batch, code/model/loss.py:57-65)."""
import torch
import torch.distributed as dist


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
