#!/usr/bin/env python
"""Device kernels of the reference-shaped eager tracking loop (bench.py dropin leg, pose_only_eager): name, launches and device time per
iteration, and the idle gaps between them (torch.profiler over 20 warmed-up iterations).  Development tool."""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from torch.profiler import profile, ProfilerActivity
    from nicer_slam_amd.tracking import TrackingStepper
    dev = torch.device("cuda", 0)
    a = argparse.Namespace(samples=128, engine="auto", precision="fp32", param_grads=True)
    model, _ = bench.make_model(a, dev)
    model.tracking_param_grads = False
    K = torch.eye(4, device=dev)
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    gen = torch.Generator(device=dev).manual_seed(1)
    batches = [bench.synth_batch(gen, 1024, dev) for _ in range(40)]
    cam = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2], device=dev)
    st = TrackingStepper(model, K[None], 1024, cam, lr=0.005, use_graph=False, world=1)
    for i in range(15):
        st.step(*batches[i])
    torch.cuda.synchronize()
    N = 20
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for i in range(N):
            st.step(*batches[15 + i])
        torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and e.time_range.end > e.time_range.start]
    ev.sort(key=lambda e: e.time_range.start)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for e in ev:
        agg[e.name[:90]][0] += e.time_range.end - e.time_range.start
        agg[e.name[:90]][1] += 1
    span = ev[-1].time_range.end - ev[0].time_range.start
    busy = sum(v[0] for v in agg.values())
    print(f"per iteration: span {span / N:.1f} us, kernels {busy / N:.1f} us, idle {(span - busy) / N:.1f} us, launches {len(ev) / N:.1f}")
    for name, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"{t / N:8.1f} us {c / N:5.1f} x  {name}")


if __name__ == "__main__":
    main()
