"""``_backend``: the three native entry points with the reference's Python-visible signatures
(code/hashencoder/src/bindings.cpp:5-7, hashencoder.h:13-15), bound to the C ABI by ctypes.

Replaces code/hashencoder/backend.py (an nvcc JIT build at import).  The tensor checks and their
messages follow code/hashencoder/src/hashencoder.cu:16-19,759-775; violations raise RuntimeError like
TORCH_CHECK does.  Launches go to torch's current HIP stream (the reference uses the legacy default
stream, which is torch's current stream unless the caller changed it).
"""
import torch

from .._native import lib, check

__all__ = ["_backend"]

def _offsets_host(offsets):
    """Host copy of the (constant) int32 offsets tensor.  Cached ON the tensor object (one D2H copy per tensor):
    a cache keyed by data_ptr would go stale when the allocator hands a freed offsets buffer to another encoder --
    wrong level geometry -> out-of-bounds gathers."""
    if not offsets.is_cuda:
        return offsets.contiguous()
    host = getattr(offsets, "_nsa_host", None)
    if host is None:
        host = offsets.detach().to("cpu", copy=True).contiguous()
        try:
            offsets._nsa_host = host
        except AttributeError:
            pass
    return host


def _dev(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")


def _flt(t, name):
    _dev(t, name)
    if t.dtype not in (torch.float32, torch.float16, torch.float64):
        raise RuntimeError(f"{name} must be a floating tensor")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: only float32 is implemented on MI355X (the reference's half/double "
                           "dispatch is never exercised by its own code)")
    return t.data_ptr()


def _int(t, name):
    _dev(t, name)
    if t.dtype != torch.int32:
        raise RuntimeError(f"{name} must be an int tensor")


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Optional per-launch timing (bench.py): set PROFILE = [] to collect (name, algorithmic_bytes, start, end) with
# events recorded on the stream the kernels are launched on (torch's current stream).
PROFILE = None


class _timed:
    def __init__(self, name, nbytes):
        self.name, self.nbytes = name, nbytes

    def __enter__(self):
        # (timing events cannot be recorded on a capturing stream: a PROFILE list left on while a consumer captures its
        # hipGraph -- fused/track_graph.py, tracking.py -- must not poison the capture)
        self.on = PROFILE is not None and not torch.cuda.is_current_stream_capturing()
        if self.on:
            self.t0 = torch.cuda.Event(enable_timing=True)
            self.t1 = torch.cuda.Event(enable_timing=True)
            self.t0.record()

    def __exit__(self, *exc):
        if self.on and PROFILE is not None:
            self.t1.record()
            PROFILE.append((self.name, self.nbytes, self.t0, self.t1))


class _Backend:
    @staticmethod
    def hash_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs, dy_dx):
        _int(offsets, "offsets")
        off = _offsets_host(offsets)
        with _timed(f"k_grid_forward<D{D},C{C},jac{int(bool(calc_grad_inputs))}>", B * L * (1 << D) * C * 4):
            check(lib.nsa_hash_encode_forward(
                _flt(inputs, "inputs"), _flt(embeddings, "embeddings"), off.data_ptr(), _flt(outputs, "outputs"),
                B, D, C, L, float(S), H, int(bool(calc_grad_inputs)), _flt(dy_dx, "dy_dx"), _stream()))

    @staticmethod
    def hash_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H,
                             calc_grad_inputs, dy_dx, grad_inputs):
        _int(offsets, "offsets")
        off = _offsets_host(offsets)
        ge = None if grad_embeddings is None else _flt(grad_embeddings, "grad_embeddings")
        check(lib.nsa_hash_encode_backward(
            _flt(grad, "grad"), _flt(inputs, "inputs"), _flt(embeddings, "embeddings"), off.data_ptr(), ge,
            B, D, C, L, float(S), H, int(bool(calc_grad_inputs)), _flt(dy_dx, "dy_dx"),
            _flt(grad_inputs, "grad_inputs"), _stream()))

    @staticmethod
    def hash_encode_second_backward(grad, inputs, embeddings, offsets, B, D, C, L, S, H, calc_grad_inputs,
                                    dy_dx, grad_grad_inputs, grad_grad, grad2_embeddings):
        _int(offsets, "offsets")
        off = _offsets_host(offsets)
        g2 = None if grad2_embeddings is None else _flt(grad2_embeddings, "grad2_embeddings")
        check(lib.nsa_hash_encode_second_backward(
            _flt(grad, "grad"), _flt(inputs, "inputs"), _flt(embeddings, "embeddings"), off.data_ptr(),
            B, D, C, L, float(S), H, int(bool(calc_grad_inputs)), _flt(dy_dx, "dy_dx"),
            _flt(grad_grad_inputs, "grad_grad_inputs"), _flt(grad_grad, "grad_grad"), g2, _stream()))


_backend = _Backend()
