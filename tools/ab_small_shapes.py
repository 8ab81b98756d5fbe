#!/usr/bin/env python
"""Small ray batches (the per-GPU share of a strong-scaling run): tracking iteration time by the sampler's tile form.
64 = two 32-point tiles per wave (the default at 1024 rays), 32 = one tile per wave, 16 = the persistent quad sampler.
usage: python tools/ab_small_shapes.py [rays ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nicer_slam_amd.fused import sampler as fs  # noqa: E402
from nicer_slam_amd.tracking import KernelTracker  # noqa: E402


def main():
    rays_list = [int(a) for a in sys.argv[1:]] or [128, 256, 512]
    dev = torch.device("cuda", 0)
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    args.samples, args.precision, args.param_grads, args.engine = 128, "fp32", False, "auto"
    model = bench.make_model(args, dev)[0]
    K = torch.eye(4, device=dev)
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    K = K[None]
    for rays in rays_list:
        for rnd in range(2):
            for tile in (64, 32, 16):
                fs.DEFAULT_TILES["sampler_small"] = fs.DEFAULT_TILES["sampler"] = tile
                model.__dict__.pop("_fused_pack", None)
                gen = torch.Generator(device=dev).manual_seed(78)
                batches = [bench.synth_batch(gen, rays, dev) for _ in range(32)]
                cam = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2], device=dev)
                tr = KernelTracker(model, K, rays, cam, lr=0.005, use_graph=True)
                for i in range(30):
                    tr.step(*batches[i % 32])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(200):
                    tr.step(*batches[i % 32])
                torch.cuda.synchronize()
                print(f"rays {rays:5d}  sampler tile {tile:2d}  {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms", flush=True)
                del tr


if __name__ == "__main__":
    main()
