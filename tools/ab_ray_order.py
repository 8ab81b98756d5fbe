#!/usr/bin/env python
"""Does the ORDER of the rays of a batch matter?  Workgroup b of the per-ray kernels (colour forward, quad SDF kernels at 128 samples per
ray) is ray b and runs on XCD b % 8 (observed placement), each XCD with its own 4 MiB L2.  Variants of the same random pixel batches:
  random        as drawn (what every caller hands over)
  morton        rays sorted along a Z-curve over the image
  xcd           the image cut into 8 regions (2 x 4), region r's rays at positions r, r + 8, r + 16, ..: one region per XCD
usage: python tools/ab_ray_order.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nicer_slam_amd.hashencoder import backend as be  # noqa: E402
from nicer_slam_amd.tracking import KernelTracker  # noqa: E402


def morton(u, v):
    def spread(x):
        x = x.long() & 0x7FF
        x = (x | (x << 8)) & 0x00FF00FF
        x = (x | (x << 4)) & 0x0F0F0F0F
        x = (x | (x << 2)) & 0x33333333
        x = (x | (x << 1)) & 0x55555555
        return x
    return spread(u) | (spread(v) << 1)


def reorder(uv, gt, how, H=680, W=1200):
    u, v = uv[0, :, 0], uv[0, :, 1]
    R = u.shape[0]
    if how == "random":
        return uv, gt
    key = morton(u, v)
    if how == "morton":
        order = torch.argsort(key)
    else:
        region = (v >= H / 2).long() * 4 + (u / (W / 4)).long().clamp(0, 3)            # 2 x 4 regions
        order = torch.argsort(region * (1 << 40) + key)
        # make the regions equally long (R / 8 each) by cutting the sorted list into 8 equal chunks, then interleave the chunks
        order = order.view(8, R // 8).t().reshape(-1)
    return uv[:, order].contiguous(), gt[order].contiguous()


def main():
    dev = torch.device("cuda", 0)
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    args.samples, args.precision, args.param_grads, args.engine = 128, "fp32", False, "auto"
    model = bench.make_model(args, dev)[0]
    K = torch.eye(4, device=dev)
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    K = K[None]
    gen = torch.Generator(device=dev).manual_seed(3)
    raw = [bench.synth_batch(gen, 1024, dev) for _ in range(40)]
    cam = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2], device=dev)
    for rnd in range(2):
        for how in ("random", "morton", "xcd"):
            batches = [reorder(uv, gt, how) for uv, gt in raw]
            tr = KernelTracker(model, K, 1024, cam, lr=0.005, use_graph=True)
            for i in range(30):
                tr.step(*batches[i % 40])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(200):
                tr.step(*batches[i % 40])
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 200 * 1e3
            eager = KernelTracker(model, K, 1024, cam, lr=0.005, use_graph=False)
            for i in range(3):
                eager.step(*batches[i])
            be.PROFILE = []
            for i in range(20):
                eager.step(*batches[i])
            torch.cuda.synchronize()
            prof, be.PROFILE = be.PROFILE, None
            agg = {}
            for name, nbytes, e0, e1 in prof:
                a = agg.setdefault(name, [0.0, 0])
                a[0] += e0.elapsed_time(e1)
                a[1] += 1
            print(f"{how:7s} {ms:.4f} ms  " + "  ".join(f"{k.replace('k_', '')} {v[0] / v[1] * 1e3:.1f}" for k, v in sorted(agg.items())), flush=True)
            del tr, eager


if __name__ == "__main__":
    main()
