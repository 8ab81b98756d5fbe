#!/usr/bin/env python
"""How full are the wave slots while k_sampler_sdf runs?  Needs the profiling build (NSA_BUILD_TAG=ts NSA_EXTRA_HIPCC_FLAGS=-DNSA_X_TS
python -m nicer_slam_amd.build; run with NSA_LIB_TAG=ts).  Every wave records its absolute start / end (s_memtime) and its hardware
slot (HW_ID: SE / SH / CU / SIMD, XCC_ID); per SIMD a sweep over those intervals gives the time with 0 / 1 / 2 / ... waves resident,
and the gap between a wave's end and the start of the next wave that takes its place.  Development tool."""
import argparse
import ctypes
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from nicer_slam_amd._native import lib
    from nicer_slam_amd.tracking import KernelTracker
    dev = torch.device("cuda", 0)
    bargs = argparse.Namespace(samples=128, engine="auto", precision="fp32", param_grads=False)
    model, conf = bench.make_model(bargs, dev)
    K = torch.eye(4, device=dev)
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    gen = torch.Generator(device=dev).manual_seed(1)
    batches = [bench.synth_batch(gen, 1024, dev) for _ in range(6)]
    cam = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2], device=dev)
    tr = KernelTracker(model, K[None], 1024, cam, use_graph=False)
    n_s = 1024 * 640 // 64
    buf = torch.zeros((n_s + 64) * 16, dtype=torch.int64, device=dev)
    lib.nsa_debug_set_ts_sampler.argtypes = [ctypes.c_void_p]
    for i in range(3):
        tr.step(*batches[i])
    torch.cuda.synchronize()
    # the sampler alone, so that k_sample_rays (same buffer) does not overwrite rows: call the sampler entry directly
    from nicer_slam_amd.fused import sampler as fs
    import nicer_slam_amd.hashencoder.backend as be
    o = torch.zeros(1024, 3, device=dev) + torch.tensor([0.1, 0.0, -0.2], device=dev)
    d = torch.nn.functional.normalize(torch.randn(1024, 3, device=dev, generator=gen), dim=-1) * 0.7
    t_rand = torch.rand(1024, 640, device=dev, generator=gen)
    for _ in range(3):
        fs.sampler_sdf(model, o, d, t_rand)
    torch.cuda.synchronize()
    assert lib.nsa_debug_set_ts_sampler(buf.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fs.sampler_sdf(model, o, d, t_rand)
    e1.record()
    torch.cuda.synchronize()
    lib.nsa_debug_set_ts_sampler(None)
    wall_us = e0.elapsed_time(e1) * 1e3
    t = buf.view(-1, 16)[:n_s].cpu()
    start, end, hw = t[:, 0], t[:, 1], t[:, 2]
    life = (end - start).double()
    xcc = (hw >> 32) & 0xF
    hwid = hw & 0xFFFFFFFF
    simd = (hwid >> 4) & 3
    cu = (hwid >> 8) & 0xF
    sh = (hwid >> 12) & 1
    se = (hwid >> 13) & 7
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    print(f"k_sampler_sdf: {n_s} waves, launch wall {wall_us:.1f} us (events, instrumented build), mean wave lifetime {life.mean():.0f} cycles "
          f"(min {life.min():.0f}, max {life.max():.0f})")
    groups = defaultdict(list)
    for i in range(n_s):
        groups[int(key[i])].append((int(start[i]), int(end[i])))
    print(f"distinct SIMDs seen: {len(groups)} (XCCs {sorted(set(xcc.tolist()))}, SEs {sorted(set(se.tolist()))}, CUs/SH {sorted(set(cu.tolist()))})")
    occ = defaultdict(float)
    spans, conc, gaps = [], [], []
    for k, iv in groups.items():
        ev = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
        t_prev, n = ev[0][0], 0
        for tt, dlt in ev:
            occ[n] += tt - t_prev
            t_prev = tt
            n += dlt
        span = ev[-1][0] - ev[0][0]
        spans.append(span)
        conc.append(sum(e - s for s, e in iv) / span)
        # hand-over gap: for every wave end (but the last two), the time until the next wave start on this SIMD at or after it
        starts = sorted(s for s, _ in iv)
        import bisect
        for _, e in iv:
            j = bisect.bisect_left(starts, e)
            if j < len(starts):
                gaps.append(starts[j] - e)
    tot = sum(occ.values())
    print("per-SIMD span (first start -> last end): mean %.0f cycles, min %.0f, max %.0f  -> %.2f GHz if the span is the launch"
          % (sum(spans) / len(spans), min(spans), max(spans), (sum(spans) / len(spans)) / wall_us / 1e3))
    print("mean resident waves per SIMD over its span: %.2f" % (sum(conc) / len(conc)))
    for n in sorted(occ):
        print("   %d waves resident: %5.1f %% of SIMD time" % (n, 100 * occ[n] / tot))
    g = torch.tensor(gaps, dtype=torch.float64)
    print("hand-over gap (a wave's end -> next wave start on the same SIMD): median %.0f, mean %.0f, p90 %.0f cycles over %d hand-overs"
          % (g.median(), g.mean(), g.quantile(0.9), len(gaps)))
    w = torch.tensor([len(v) for v in groups.values()], dtype=torch.float64)
    print("waves per SIMD: mean %.1f, min %d, max %d" % (w.mean(), w.min(), w.max()))


if __name__ == "__main__":
    main()
