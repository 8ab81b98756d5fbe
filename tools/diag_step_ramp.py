"""Where does a SHORT timed region lose its time?  bench.py's tracker line costs 0.566 ms per iteration over 200 iterations and
0.583 over the driver's 20 (same box, same binaries, whatever the pre-warm recipe: profiles/r06_ab_experiments.txt r6p) -- a fixed
~0.35 ms per timed region.  This script replays bench.py's region (fence, K graph iterations, fence) with an event after every
iteration and host clocks around every call, and prints where the region's time goes.

    python tools/diag_step_ramp.py [--steps 20] [--idle-ms 0]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--idle-ms", type=float, default=0.0, help="host sleep between the opening fence and the first iteration")
    ap.add_argument("--fresh", action="store_true", help="as bench.py: every warm-up / timed batch is one the device has not rendered before")
    ap.add_argument("--reset", action="store_true", help="as bench.py: stepper.reset() between the pre-warm iterations and the warm-up")
    ap.add_argument("--gemm-s", type=float, default=0.0, help="as bench.py: seconds of library fp32 GEMMs first")
    ap.add_argument("--gc-off", action="store_true")
    a = ap.parse_args()
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model, conf = bench.make_model(args, device)
    K = torch.eye(4, device=device)
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    K = K[None]
    gen = torch.Generator(device=device).manual_seed(1)
    n = a.warmup + a.steps
    batches = [bench.synth_batch(gen, args.rays, device) for _ in range(n * (a.rounds if a.fresh else 1))]
    gen_pre = torch.Generator(device=device).manual_seed(1000)
    pre = [bench.synth_batch(gen_pre, args.rays, device) for _ in range(32)] if a.fresh else batches
    cam = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2], device=device)
    from nicer_slam_amd.tracking import KernelTracker
    st = KernelTracker(model, K, args.rays, cam, lr=0.005, use_graph=True, world=1)
    if a.gc_off:
        import gc
        gc.collect()
        gc.disable()
    if a.gemm_s:
        x = torch.randn(4096, 4096, device=device)
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < a.gemm_s:
            for _ in range(8):
                x @ x
            torch.cuda.synchronize()
        del x
    cam0 = st.cam.detach().clone()
    out = []
    for rnd in range(a.rounds):
        off = rnd * n if a.fresh else 0
        if rnd == 0 or a.fresh:
            for i in range(100):
                st.step(*pre[i % len(pre)])
            if a.reset:
                st.reset(cam0)
        for i in range(a.warmup):
            st.step(*batches[off + i])
        torch.cuda.synchronize()
        if a.idle_ms:
            time.sleep(a.idle_ms * 1e-3)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        host = []
        for i in range(a.steps):
            h0 = time.perf_counter()
            st.step(*batches[off + a.warmup + i])
            ev[i + 1].record()
            host.append((time.perf_counter() - h0) * 1e6)
        t_queued = time.perf_counter()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        gpu = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(a.steps)]
        out.append({"wall_ms_per_step": round((t1 - t0) / a.steps * 1e3, 4),
                    "queued_after_us": round((t_queued - t0) * 1e6, 1), "region_us": round((t1 - t0) * 1e6, 1),
                    "gpu_first_to_last_event_us": round(ev[0].elapsed_time(ev[-1]) * 1e3, 1),
                    "gpu_step_us": [round(x, 1) for x in gpu], "host_step_us": [round(x, 1) for x in host]})
    print(json.dumps({"steps": a.steps, "idle_ms": a.idle_ms, "fresh": a.fresh, "reset": a.reset, "gemm_s": a.gemm_s, "rounds": out}))


if __name__ == "__main__":
    main()
