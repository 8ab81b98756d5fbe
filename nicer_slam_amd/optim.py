"""torch.optim.Adam replacement for the mapping optimizer (reference code/training/volsdf_train.py:150-174:
``torch.optim.Adam(para_list, betas=(0.9, 0.99), eps=1e-15)`` over the three grid tables -- one of them 1 GiB -- and
the two small MLPs).  One HIP pass per parameter tensor (csrc/map_tail.hip: 4 reads + 3 writes per element, the HBM
floor of a dense Adam step) instead of torch's seven multi-tensor passes.

``consume_table_grads`` (default on): a grid table whose ``.grad`` is the fused mapping engine's persistent buffer
(fused/tablegrad.py) has that gradient CONSUMED by ``step()`` -- once read it is zero-filled on a side stream, underneath the
next forward pass, so that the next backward finds a clean buffer without a fill in its way; such a ``.grad`` must not be read
after ``step()`` (it is being cleared; zero once the next backward has started).  ``"fused"`` clears inside the step kernel
instead (zero right after ``step()``, but a slower step).  Every other gradient is left untouched, as torch does.

Same semantics, operation order and state layout as torch.optim.Adam without weight decay / amsgrad / maximize:
``state[p] = {"step": tensor(float), "exp_avg", "exp_avg_sq"}``, so state_dicts are interchangeable.  CUDA float32
contiguous parameters only; anything else raises (no fallback).
"""
import torch

from ._native import lib, check
from ._version import bump_version
from .fused import tablegrad



class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, consume_table_grads=True):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.consume_table_grads = consume_table_grads

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        st = torch.cuda.current_stream().cuda_stream
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.dtype == torch.float32):
                    raise RuntimeError("nicer_slam_amd.optim.Adam: float32 contiguous CUDA parameters only")
                if g.is_sparse:
                    raise RuntimeError("nicer_slam_amd.optim.Adam does not support sparse gradients")
                g = g.contiguous()
                state = self.state[p]
                if not state:
                    state["step"] = torch.zeros((), dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                # a table gradient living in the fused engine's persistent buffer is consumed: read and left zero, so the next
                # backward scatters into it without a fill (fused/tablegrad.py)
                consume = self.consume_table_grads if tablegrad.consumable(p, g) else False
                step_fn = lib.nsa_adam_table_step_clear if consume == "fused" else lib.nsa_adam_table_step
                check(step_fn(p.data_ptr(), g.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(), p.numel(),
                              int(state["step"]), float(group["lr"]), float(b1), float(b2), float(group["eps"]), st))
                if consume == "fused":
                    tablegrad.mark_clean(p)
                elif consume:
                    tablegrad.clear_async(p)
                # the kernel wrote p behind autograd's back: bump its version counter like an in-place op would
                # (the packed-weight caches of the fused engine key on it)
                bump_version(p)
        return loss
