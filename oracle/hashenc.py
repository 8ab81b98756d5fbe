"""oracle/hashenc.py -- TEST INFRASTRUCTURE ONLY (never on the product path).

ctypes front-end of ``oracle/hashenc_oracle.c`` (the CPU restatement of
``code/hashencoder/src/hashencoder.cu``).  :class:`OracleBackend` exposes the three
functions with the calling convention of the reference's native module
(``code/hashencoder/src/bindings.cpp:5-7`` / ``hashencoder.h:13-15``) on **CPU** torch
tensors, so the reference's own Python wrappers (``code/hashencoder/hashgrid.py:13-134``)
can be executed on top of it to capture golden vectors, and so tests can compare the HIP
path with it.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnso_oracle.so")
_lib = None


def build(force=False):
    """Compile the C oracle with gcc (``make -C oracle``)."""
    src = os.path.join(_HERE, "hashenc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnso_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        u32, f32, vp, i32 = ctypes.c_uint32, ctypes.c_float, ctypes.c_void_p, ctypes.c_int
        _lib.nso_set_threads.restype = None
        _lib.nso_set_threads.argtypes = [i32]
        _lib.nso_set_threads(min(16, os.cpu_count() or 1))
        _lib.nso_fast_hash.restype = u32
        _lib.nso_fast_hash.argtypes = [vp, u32]
        _lib.nso_level_row.restype = u32
        _lib.nso_level_row.argtypes = [u32, u32, vp, u32]
        _lib.nso_level_geometry.restype = None
        _lib.nso_level_geometry.argtypes = [vp, u32, f32, u32, vp, vp, vp, vp]
        _lib.nso_hash_encode_forward.restype = i32
        _lib.nso_hash_encode_forward.argtypes = [vp, vp, vp, vp, u32, u32, u32, u32, f32, u32, i32, vp]
        _lib.nso_hash_encode_backward.restype = i32
        _lib.nso_hash_encode_backward.argtypes = [vp, vp, vp, vp, vp, u32, u32, u32, u32, f32, u32,
                                                  i32, vp, vp]
        _lib.nso_hash_encode_second_backward.restype = i32
        _lib.nso_hash_encode_second_backward.argtypes = [vp, vp, vp, vp, u32, u32, u32, u32, f32, u32,
                                                         i32, vp, vp, vp, vp]
    return _lib


def fast_hash(cell):
    a = np.ascontiguousarray(cell, dtype=np.uint32)
    return int(lib().nso_fast_hash(a.ctypes.data, a.size))


def level_row(rows, res, cell):
    a = np.ascontiguousarray(cell, dtype=np.uint32)
    return int(lib().nso_level_row(rows, res, a.ctypes.data, a.size))


def level_geometry(offsets, level, S, H):
    """(row0, rows, resolution, scale) of one level, float32 arithmetic as the kernel."""
    off = np.ascontiguousarray(offsets, dtype=np.int32)
    row0, rows, res = (ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32())
    scale = ctypes.c_float()
    lib().nso_level_geometry(off.ctypes.data, level, float(S), H, ctypes.byref(row0),
                             ctypes.byref(rows), ctypes.byref(res), ctypes.byref(scale))
    return row0.value, rows.value, res.value, scale.value


def _chk(t, name, dtype=torch.float32):
    # mirrors the reference's TORCH_CHECKs (hashencoder.cu:16-19, 759-775) minus "is CUDA"
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be a {'int' if dtype == torch.int32 else 'floating'} tensor")
    if t.device.type != "cpu":
        raise RuntimeError(f"{name}: the oracle only takes CPU tensors")
    return t.data_ptr()


_ERR = "GridEncoding: C must be 1, 2, 4, or 8."


class OracleBackend:
    """CPU stand-in with the interface of ``hashencoder.backend._backend``."""

    @staticmethod
    def hash_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H,
                            calc_grad_inputs, dy_dx):
        rc = lib().nso_hash_encode_forward(
            _chk(inputs, "inputs"), _chk(embeddings, "embeddings"),
            _chk(offsets, "offsets", torch.int32), _chk(outputs, "outputs"),
            B, D, C, L, float(S), H, int(bool(calc_grad_inputs)), _chk(dy_dx, "dy_dx"))
        if rc:
            raise RuntimeError(_ERR)

    @staticmethod
    def hash_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H,
                             calc_grad_inputs, dy_dx, grad_inputs):
        rc = lib().nso_hash_encode_backward(
            _chk(grad, "grad"), _chk(inputs, "inputs"), _chk(embeddings, "embeddings"),
            _chk(offsets, "offsets", torch.int32), _chk(grad_embeddings, "grad_embeddings"),
            B, D, C, L, float(S), H, int(bool(calc_grad_inputs)), _chk(dy_dx, "dy_dx"),
            _chk(grad_inputs, "grad_inputs"))
        if rc:
            raise RuntimeError(_ERR)

    @staticmethod
    def hash_encode_second_backward(grad, inputs, embeddings, offsets, B, D, C, L, S, H,
                                    calc_grad_inputs, dy_dx, grad_grad_inputs, grad_grad,
                                    grad2_embeddings):
        rc = lib().nso_hash_encode_second_backward(
            _chk(grad, "grad"), _chk(inputs, "inputs"), _chk(embeddings, "embeddings"),
            _chk(offsets, "offsets", torch.int32), B, D, C, L, float(S), H,
            int(bool(calc_grad_inputs)), _chk(dy_dx, "dy_dx"),
            _chk(grad_grad_inputs, "grad_grad_inputs"), _chk(grad_grad, "grad_grad"),
            _chk(grad2_embeddings, "grad2_embeddings"))
        if rc:
            raise RuntimeError(_ERR)
