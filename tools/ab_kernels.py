#!/usr/bin/env python
"""A/B timing of the tracking iteration's kernels for one build of the library (NSA_LIB_TAG selects an experiment build made
with NSA_BUILD_TAG / NSA_EXTRA_HIPCC_FLAGS, see nicer_slam_amd/build.py).  Prints one JSON line: per-kernel average launch
duration (events on the launch stream, eager replay of the bench batches) and the graph-replayed ms/iteration.

    NSA_LIB_TAG=nosplit python tools/ab_kernels.py [--steps 40]
Development tool.  (The wrong-numbers ablation macros NSA_ABL_* of rounds 1-3 were removed from the kernel headers in round 4; their
results are in profiles/r0[1-3]_ab_experiments.txt, the code in the history up to commit 2d72c93.)"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--no-graph-leg", action="store_true")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16_colour"])
    args = ap.parse_args()
    import bench
    from nicer_slam_amd.hashencoder import backend as be
    from nicer_slam_amd.tracking import KernelTracker
    dev = torch.device("cuda", 0)
    bargs = argparse.Namespace(samples=args.samples, engine="auto", precision=args.precision, param_grads=False)
    model, conf = bench.make_model(bargs, dev)
    K = torch.eye(4, device=dev)
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    gen = torch.Generator(device=dev).manual_seed(1)
    batches = [bench.synth_batch(gen, args.rays, dev) for _ in range(args.steps)]
    cam = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2], device=dev)
    a = torch.randn(4096, 4096, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        for _ in range(8):
            a @ a
        torch.cuda.synchronize()
    out = {"tag": os.environ.get("NSA_LIB_TAG", ""), "precision": args.precision, "rays": args.rays, "samples": args.samples,
           "env": {k: v for k, v in os.environ.items() if k.startswith("NSA_") and k != "NSA_LIB_TAG"}}
    if not args.no_graph_leg:
        tr = KernelTracker(model, K[None], args.rays, cam, use_graph=True)
        for i in range(10):
            tr.step(*batches[i % len(batches)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            tr.step(*batches[i])
        torch.cuda.synchronize()
        out["graph_ms"] = round((time.perf_counter() - t0) / args.steps * 1e3, 4)
    eager = KernelTracker(model, K[None], args.rays, cam, use_graph=False)
    for i in range(5):
        eager.step(*batches[i])
    be.PROFILE = []
    for i in range(args.steps):
        eager.step(*batches[i])
    torch.cuda.synchronize()
    prof, be.PROFILE = be.PROFILE, None
    agg = {}
    for name, nbytes, e0, e1 in prof:
        a_ = agg.setdefault(name, [0.0, 0])
        a_[0] += e0.elapsed_time(e1)
        a_[1] += 1
    out["kernels_us"] = {k: round(v[0] / v[1] * 1e3, 1) for k, v in sorted(agg.items())}
    out["sum_us"] = round(sum(out["kernels_us"].values()), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
