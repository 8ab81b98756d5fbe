"""Side-stream work on parameter tensors and the events main-stream consumers wait for.

Two users: the deferred zero fill of a table-gradient buffer (fused/tablegrad.py, policy "async") and the overlapped Adam step of
a large table (optim.Adam(overlap_min_numel=...)): the 1 GiB colour table's step is pure HBM streaming and its next reader is the
colour forward, ~2.5 ms into the next iteration -- behind the ray sampler and the SDF forward, which do not touch it.  The step is
issued on this module's side stream; every fused-engine consumer of a table (sampler.grid_desc, the hash encoder module, the
gradient buffer's acquisition) calls settle(tensor) first, which makes the CURRENT stream wait for the pending event.  Code outside
the engine that reads such a parameter (checkpointing, torch ops) must call settle_all() / optimizer.synchronize() first."""
import torch
from torch.utils.weak import WeakTensorKeyDictionary

_pending = WeakTensorKeyDictionary()
_side = {}


def side_stream(device):
    s = _side.get(device)
    if s is None:
        s = _side[device] = torch.cuda.Stream(device=device)
    return s


def defer(tensor, event):
    _pending[tensor] = event


def settle(tensor):
    """the current stream waits for side-stream work pending on ``tensor`` (no-op when there is none)"""
    if len(_pending) == 0:
        return
    ev = _pending.pop(tensor, None)
    if ev is not None:
        torch.cuda.current_stream(tensor.device).wait_event(ev)


def settle_all():
    for t in list(_pending.keys()):
        settle(t)
