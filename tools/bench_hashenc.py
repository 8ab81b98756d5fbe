"""Time the stand-alone grid-encoder operator (C ABI section 1 = the reference's hashencoder extension) on the three shipped
grid geometries: forward (+ Jacobian), first backward (table scatter + J^T g), second backward.
usage: python tools/bench_hashenc.py [B]      (NSA_LIB_TAG selects an A/B build of the library)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from nicer_slam_amd.hashencoder.hashgrid import HashEncoder
from nicer_slam_amd.hashencoder.backend import _backend

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192 * 98
dev = "cuda"
torch.manual_seed(0)
GEOMS = {"colour L16 C2 (1 GiB)": dict(num_levels=16, level_dim=2, base_resolution=16, desired_resolution=2048, log2_hashmap_size=24),
         "fine   L8  C4 (36 MiB)": dict(num_levels=8, level_dim=4, base_resolution=32, desired_resolution=128, log2_hashmap_size=19),
         "coarse L4  C8 (4 MiB)": dict(num_levels=4, level_dim=8, base_resolution=32, desired_resolution=32, log2_hashmap_size=19)}


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n * 1e3


# points as the renderer produces them: samples along rays through the unit cube (consecutive points = one ray)
R, S = B // 98, 98
o = torch.rand(R, 1, 3, device=dev) * 0.2 + 0.4
d = torch.nn.functional.normalize(torch.randn(R, 1, 3, device=dev), dim=-1)
x = (o + d * torch.linspace(0.0, 0.45, S, device=dev).view(1, S, 1)).reshape(-1, 3).clamp(0, 1).contiguous()
B = x.shape[0]
for name, kw in GEOMS.items():
    enc = HashEncoder(input_dim=3, **kw).to(dev)
    enc.embeddings.data.uniform_(-0.1, 0.1)
    L, C, D = enc.num_levels, enc.level_dim, 3
    S_, H = float(np.log2(enc.per_level_scale)), enc.base_resolution
    out = torch.empty(L, B, C, device=dev)
    dy = torch.empty(B, L * D * C, device=dev)
    g = torch.randn(L, B, C, device=dev)
    ge = torch.zeros_like(enc.embeddings)
    gi = torch.zeros(B, 3, device=dev)
    ggi = torch.randn(B, 3, device=dev)
    gg = torch.zeros(L, B, C, device=dev)
    g2 = torch.zeros_like(enc.embeddings)
    e, off = enc.embeddings.data, enc.offsets
    gather = B * L * 8 * C * 4
    rows = [("forward", lambda: _backend.hash_encode_forward(x, e, off, out, B, D, C, L, S_, H, False, dy), gather + B * L * C * 4 + B * 12),
            ("forward + Jacobian", lambda: _backend.hash_encode_forward(x, e, off, out, B, D, C, L, S_, H, True, dy),
             gather + B * L * C * 4 * (1 + D) + B * 12),
            ("backward (scatter + J^T g)", lambda: _backend.hash_encode_backward(g, x, e, off, ge, B, D, C, L, S_, H, True, dy, gi),
             gather + B * L * C * 4 * (1 + D) + B * 24),
            ("second backward", lambda: _backend.hash_encode_second_backward(g, x, e, off, B, D, C, L, S_, H, True, dy, ggi, gg, g2),
             gather + B * L * C * 4 * (2 + D) + B * 24)]
    for what, fn, nbytes in rows:
        us = timed(fn)
        print(f"{name:24s} {what:28s} B={B}: {us:8.1f} us   {nbytes / us / 1e6:6.2f} TB/s algorithmic")
