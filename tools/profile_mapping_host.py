#!/usr/bin/env python
"""Which Python lines launch the GPU kernels of one MAPPING iteration (bench.py::mapping_leg) that are NOT the engine's own (nsa::)?
torch.profiler with stacks over 3 warmed-up iterations; GPU time and launch count per (kernel, innermost frame inside this repository).
Development tool (what is left of the mapping step outside the HIP library)."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from torch.profiler import profile, ProfilerActivity
    dev = torch.device("cuda", 0)
    N = 3

    def hook(step):
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            for _ in range(N):
                step()
            torch.cuda.synchronize()
        return prof

    # (1) who calls what: a dispatch mode sees every aten op of the forward thread with the Python stack that issued it
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    calls = collections.defaultdict(int)

    class Who(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = func.__name__ if hasattr(func, "__name__") else str(func)
            frame = "?"
            for fs_ in reversed(traceback.extract_stack()[:-1]):
                if ROOT in fs_.filename and "profile_mapping_host" not in fs_.filename:
                    frame = "%s:%d" % (fs_.filename.replace(ROOT + "/", ""), fs_.lineno)
                    break
            calls[(str(func), frame)] += 1
            return func(*args, **(kwargs or {}))

    def hook_who(step):
        with Who():
            step()
        torch.cuda.synchronize()
        return None

    bench.mapping_leg(dev, iters=1, cpu=False, step_hook=hook_who)
    print("-- aten ops of one iteration's forward thread, by issuing line (count) --")
    for (op, frame), c in sorted(calls.items(), key=lambda kv: -kv[1])[:70]:
        print("%5d  %-44s %s" % (c, op[:44], frame))
    prof = bench.mapping_leg(dev, iters=2, cpu=False, step_hook=hook)
    ev = prof.events()
    # kernel events carry no stack: map them through their launching CPU op (correlation id -> enclosing op with a stack)
    by = collections.defaultdict(lambda: [0.0, 0])
    total = collections.defaultdict(float)
    for e in ev:
        if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
            continue
        frame = "?"
        for fr in (e.stack or []):
            if ROOT in fr and "profile_mapping_host" not in fr:
                frame = fr.replace(ROOT + "/", "")
                break
        for k in e.kernels:
            nm = k.name
            total["nsa" if "nsa::" in nm else "other"] += k.duration
            if "nsa::" in nm:
                continue
            key = (nm[:70], frame[:90], e.name[:40])
            by[key][0] += k.duration
            by[key][1] += 1
    # (2) the device timeline: busy time, idle gaps and what surrounds the largest ones
    dk = sorted([e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA and e.time_range.end > e.time_range.start],
                key=lambda e: e.time_range.start)
    if dk:
        span = dk[-1].time_range.end - dk[0].time_range.start
        busy = sum(e.time_range.end - e.time_range.start for e in dk)
        gaps = []
        end = dk[0].time_range.end
        prev = dk[0]
        for e in dk[1:]:
            if e.time_range.start > end:
                gaps.append((e.time_range.start - end, prev.name[:60], e.name[:60]))
            if e.time_range.end > end:
                end, prev = e.time_range.end, e
        idle = sum(g[0] for g in gaps)
        print("\n-- device timeline over %d iterations: span %.2f ms, busy %.2f ms (sum of kernels), idle %.2f ms in %d gaps --"
              % (N, span / 1e3, busy / 1e3, idle / 1e3, len(gaps)))
        print("   per iteration: span %.2f ms, idle %.2f ms; gaps > 20 us: %d (%.2f ms)"
              % (span / 1e3 / N, idle / 1e3 / N, sum(1 for g in gaps if g[0] > 20) / N, sum(g[0] for g in gaps if g[0] > 20) / 1e3 / N))
        agg_gap = collections.defaultdict(lambda: [0.0, 0])
        for g, a, b in gaps:
            agg_gap[(a, b)][0] += g
            agg_gap[(a, b)][1] += 1
        print("   largest gap classes (us per iteration, count per iteration, after -> before):")
        for (a, b), (t, c) in sorted(agg_gap.items(), key=lambda kv: -kv[1][0])[:25]:
            print("%8.1f %5.1f  %s  ->  %s" % (t / N, c / N, a, b))
    print("per iteration: engine kernels %.1f us, other kernels %.1f us" % (total["nsa"] / N, total["other"] / N))
    agg_line = collections.defaultdict(lambda: [0.0, 0])
    for (nm, frame, op), (t, c) in by.items():
        agg_line[frame][0] += t
        agg_line[frame][1] += c
    print("\n-- by source line (us / iteration, launches / iteration) --")
    for frame, (t, c) in sorted(agg_line.items(), key=lambda kv: -kv[1][0])[:40]:
        print("%8.1f %6.1f  %s" % (t / N, c / N, frame))
    print("\n-- by kernel x line --")
    for (nm, frame, op), (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:50]:
        print("%8.1f %6.1f  %-40s %-70s %s" % (t / N, c / N, op, nm, frame))


if __name__ == "__main__":
    main()
