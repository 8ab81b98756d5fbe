"""Version-counter bump for parameters written by a kernel behind autograd's back (the packed-weight caches of the fused engine
key on ``(data_ptr, _version)``, fused/sampler.py::packed_sdf, fused/render.py::packed_colour)."""
import torch


def bump_version(p):
    """Advance p's autograd version counter like an in-place op would."""
    try:
        torch._C._autograd._unsafe_set_version_counter((p,), (p._version + 1,))
    except (AttributeError, TypeError):     # torch without that private hook (or with another signature): a no-op in-place
        p.add_(0)                          # op does the same (one extra pass over p; only MLP-sized tensors are cached on it)


def _self_check():
    t = torch.zeros(2)
    v = t._version
    bump_version(t)
    if t._version <= v:
        raise ImportError("nicer_slam_amd: cannot advance tensor version counters on this torch build -- the fused engine's "
                          "packed-weight caches would go stale after an optimizer step")


_self_check()
