"""Persistent, in-place gradient buffers of the hash-grid tables for the fused mapping engine.

The reference's mapping iteration (code/training/volsdf_train.py:547-576) runs optimizer.zero_grad(), a backward whose grid
encoder allocates a zero-filled dense gradient per table and atomically adds into it (code/hashencoder/hashgrid.py:85,117-118),
and a dense Adam step.  With a 1 GiB colour table the zero fill and autograd's sum of the two passes' table gradients (composite
pass + eikonal pass) are pure HBM streaming.  Here every table owns ONE persistent gradient buffer:

* the MAP backward kernels scatter straight into ``param.grad`` -- which is this buffer whenever ``param.grad`` was None (the
  state optimizer.zero_grad() leaves) -- so both passes of an iteration accumulate in place and autograd has nothing to add;
* the buffer is zero-filled by the engine right before the first MAP kernel of a backward pass adds into it (nsa_fill_zero on
  the launch stream).  Alternatives measured and not kept (profiles/r05_ab_experiments.txt r5w-r5z): clearing inside the Adam
  kernel (kept as an option, nsa_adam_table_step_clear: the eighth stream slows the 1 GiB step by more than the fill costs), the
  fill on a side stream underneath the next forward pass, and the 1 GiB table's Adam step itself on a side stream underneath the
  next iteration's ray sampler (no gain either: the forward kernels slow down by what the overlap hides).

Observable semantics: after ``loss.backward()`` ``param.grad`` holds the accumulated gradient exactly as with autograd
(including accumulation over several backward calls, and into a ``.grad`` tensor the caller put there).  Differences
(INTEGRATION.md): the table gradients do not travel through autograd (``torch.autograd.grad(loss, table)`` raises;
tensor hooks on the tables do not fire), the gradient tensor is the SAME storage every iteration (a ``.grad`` kept across
iterations is overwritten by the next backward), and with ``optim.Adam(consume_table_grads=True)`` a table's ``.grad`` reads
zero after ``step()``.  ``NSA_TABLE_GRADS=autograd`` (or ``IN_PLACE = False``) restores fresh
zero-filled gradients returned through autograd.
"""
import os

import torch
from torch.utils.weak import WeakTensorKeyDictionary

IN_PLACE = os.environ.get("NSA_TABLE_GRADS", "inplace") != "autograd"


class _Entry:
    __slots__ = ("buf", "clean", "clean_version")

    def __init__(self, param):
        self.buf = torch.zeros_like(param, memory_format=torch.contiguous_format)
        self.clean = True
        self.clean_version = self.buf._version      # the buffer's autograd version counter when it was last known to be all zero


def _fill(buf, stream):
    from .._native import lib, check
    check(lib.nsa_fill_zero(buf.data_ptr(), buf.numel(), stream.cuda_stream))


_pool = WeakTensorKeyDictionary()


def _entry(param):
    e = _pool.get(param)
    if e is None or e.buf.shape != param.shape or e.buf.device != param.device:
        e = _pool[param] = _Entry(param)
    return e


def target(param):
    """The tensor the MAP kernels of this backward pass add ``param``'s gradient into (float32, contiguous, param's shape);
    afterwards it is (part of) ``param.grad``.  Called inside autograd.Function.backward; the Function returns None for the table."""
    if not (param.is_cuda and param.dtype == torch.float32):
        raise RuntimeError("fused mapping engine: grid tables must be float32 CUDA tensors")
    g = param.grad
    if g is not None:
        if g.dtype == torch.float32 and g.is_contiguous() and g.shape == param.shape and g.device == param.device and not g.is_sparse:
            e = _pool.get(param)
            if e is not None and e.buf.data_ptr() == g.data_ptr():
                e.clean = False
            return g                       # accumulate where the gradient already lives (ours or the caller's)
        raise RuntimeError("fused mapping engine: a table's .grad must be a dense contiguous float32 tensor of the table's shape "
                           "(set NSA_TABLE_GRADS=autograd for gradients returned through autograd)")
    e = _entry(param)
    # holds the previous pass's gradient unless new or cleared by the optimizer (consume_table_grads) -- and, in that case, only if no
    # torch op wrote the buffer since (it stays reachable as param.grad after the step: AccumulateGrad's `+=` of the composed engine
    # or a caller's in-place add bump its version counter; the library's own kernels go through target(), which un-cleans)
    if not e.clean or e.buf._version != e.clean_version:
        with torch.cuda.device(e.buf.device):
            _fill(e.buf, torch.cuda.current_stream())
    e.clean = False
    with torch.no_grad():
        param.grad = e.buf
    return e.buf


def consumable(param, grad):
    """True when ``grad`` is this module's buffer of ``param`` (the optimizer may then consume it: read it and leave it zero)."""
    e = _pool.get(param)
    return e is not None and grad.data_ptr() == e.buf.data_ptr()


def mark_clean(param):
    """the optimizer cleared the buffer itself (nsa_adam_table_step_clear)"""
    e = _pool.get(param)
    if e is not None:
        e.clean = True
        e.clean_version = e.buf._version
