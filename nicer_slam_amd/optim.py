"""torch.optim.Adam replacement for the mapping optimizer (reference code/training/volsdf_train.py:150-174:
``torch.optim.Adam(para_list, betas=(0.9, 0.99), eps=1e-15)`` over the three grid tables -- one of them 1 GiB -- and
the two small MLPs).  One HIP pass per parameter tensor (csrc/map_tail.hip: 4 reads + 3 writes per element, the HBM
floor of a dense Adam step) instead of torch's seven multi-tensor passes.

``consume_table_grads``: what happens to a grid table's gradient that lives in the fused mapping engine's persistent buffer
(fused/tablegrad.py) once ``step()`` has read it.  ``False`` (default): nothing -- the engine zero-fills the buffer right before
the next backward scatters into it (nsa_fill_zero).  ``True``: cleared inside the step kernel (nsa_adam_table_step_clear; such a
``.grad`` reads zero after ``step()``) -- measured SLOWER on MI355X: an eighth concurrent stream slows the 1 GiB step by more than
the separate fill costs (profiles/r05_ab_experiments.txt r5w).  Every other gradient is left untouched, as torch does.
Tensors of up to 65536 elements (the MLP parameters) are stepped 24 per launch (nsa_adam_multi_step).

``none_grad``: what ``step()`` does with a parameter whose ``.grad`` is None.  The reference was written for, and its results were
produced under, torch 1.11 (env_yamls/nicer-slam.yaml:62): there ``optimizer.zero_grad()`` (volsdf_train.py:547) leaves ZERO tensors, so a
table that receives no gradient in ``stage="coarse"`` (first 25 % of a mapping round) or ``color_stage="base"`` (first 70 %, :550-555) is
still stepped -- both moments decay and the parameter keeps moving along its momentum.  Under torch >= 2.0 the same unmodified line sets
``.grad = None`` and torch.optim.Adam skips such a parameter: a different map for 25-70 % of every mapping round.
``"skip"`` (default) = the installed torch's semantics (what ``torch.optim.Adam`` does on this box);
``"zeros"`` = the reference environment's: a parameter with ``.grad is None`` that HAS optimizer state (i.e. was stepped before -- under
torch 1.11 its ``.grad`` would be a zero tensor from then on) takes a zero-gradient step (nsa_adam_table_step_zero_grad: no gradient is
read, no 1 GiB of zeros is written or streamed); one that has never been stepped is skipped as torch 1.11 skips a parameter whose
``.grad`` was never populated.  Equal to ``torch.optim.Adam`` driven with ``zero_grad(set_to_none=False)`` (tests/test_mapping_gpu.py).

Same semantics, operation order and state layout as torch.optim.Adam without weight decay / amsgrad / maximize:
``state[p] = {"step": tensor(float), "exp_avg", "exp_avg_sq"}``, so state_dicts are interchangeable.  CUDA float32
contiguous parameters only; anything else raises (no fallback).
"""
import torch

from ._native import lib, check, AdamSeg
from ._version import bump_version
from .fused import tablegrad



class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, consume_table_grads=False, none_grad="skip"):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        if none_grad not in ("skip", "zeros"):
            raise ValueError(f"none_grad={none_grad!r}: expected 'skip' (installed torch's semantics) or 'zeros' (torch 1.11's)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.consume_table_grads = bool(consume_table_grads)
        self.none_grad = none_grad
        self._zeros = {}

    def __getstate__(self):
        state = super().__getstate__()          # (torch keeps defaults / state / param_groups only)
        state.update(consume_table_grads=self.consume_table_grads, none_grad=self.none_grad)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__.setdefault("consume_table_grads", False)      # (pickles made before these options existed)
        self.__dict__.setdefault("none_grad", "skip")
        self.__dict__.setdefault("_zeros", {})

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # a torch state_dict saved with capturable / fused Adam holds `step` on the device: int(step) in step() would then
        # synchronise once per tensor per iteration -- move it to the host once
        for st in self.state.values():
            if torch.is_tensor(st.get("step")) and st["step"].device.type != "cpu":
                st["step"] = st["step"].detach().to("cpu", torch.float32)

    def _zero_grad_of(self, p):
        """a zero gradient for a SMALL parameter under none_grad="zeros" (shared, read-only, <= 256 KiB per device)"""
        z = self._zeros.get(p.device)
        if z is None:
            z = self._zeros[p.device] = torch.zeros(self.SMALL, dtype=torch.float32, device=p.device)
        return z

    SMALL = 1 << 16        # tensors up to this many elements share launches (nsa_adam_multi_step, 24 per launch)

    def _flush(self, batch, key, st):
        if not batch:
            return
        b1, b2, eps = key[1:]
        segs = (AdamSeg * len(batch))(*[AdamSeg(p.data_ptr(), g.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr(),
                                                p.numel(), int(s["step"]), lr) for p, g, s, lr in batch])
        check(lib.nsa_adam_multi_step(segs, len(batch), b1, b2, eps, st))
        for p, _, _, _ in batch:
            bump_version(p)
        batch.clear()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        st = torch.cuda.current_stream().cuda_stream
        batch, batch_key = [], None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                g = p.grad
                if g is None:
                    # torch >= 2.0: skipped.  torch 1.11 (none_grad="zeros"): a parameter stepped before holds a zero .grad
                    if self.none_grad != "zeros" or not self.state.get(p):
                        continue
                    if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                        raise RuntimeError("nicer_slam_amd.optim.Adam: float32 contiguous CUDA parameters only")
                    state = self.state[p]
                    state["step"] += 1
                    if 0 < p.numel() <= self.SMALL:
                        key = (p.device, float(b1), float(b2), float(group["eps"]))
                        if key != batch_key or len(batch) == 24:
                            self._flush(batch, batch_key, st)
                            batch_key = key
                        batch.append((p, self._zero_grad_of(p), state, float(group["lr"])))
                        continue
                    check(lib.nsa_adam_table_step_zero_grad(p.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(),
                                                            p.numel(), int(state["step"]), float(group["lr"]), float(b1), float(b2),
                                                            float(group["eps"]), st))
                    bump_version(p)
                    continue
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
                if 0 < p.numel() <= self.SMALL:
                    key = (p.device, float(b1), float(b2), float(group["eps"]))
                    if key != batch_key or len(batch) == 24:
                        self._flush(batch, batch_key, st)
                        batch_key = key
                    batch.append((p, g, state, float(group["lr"])))      # g: kept alive until the launch
                    continue
                # a table gradient living in the fused engine's persistent buffer can be consumed: zero-filled behind the read
                # (fused/tablegrad.py)
                consume = bool(getattr(self, "consume_table_grads", False)) and tablegrad.consumable(p, g)
                step_fn = lib.nsa_adam_table_step_clear if consume else lib.nsa_adam_table_step
                check(step_fn(p.data_ptr(), g.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(), p.numel(),
                              int(state["step"]), float(group["lr"]), float(b1), float(b2), float(group["eps"]), st))
                if consume:
                    tablegrad.mark_clean(p)
                # the kernel wrote p behind autograd's back: bump its version counter like an in-place op would
                # (the packed-weight caches of the fused engine key on it)
                bump_version(p)
        self._flush(batch, batch_key, st)
        return loss
