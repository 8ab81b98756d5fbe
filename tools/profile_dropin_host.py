"""Host-time breakdown of the reference's tracking loop shape (volsdf_train.py:406-443) on the fused engine, eager:
camera -> get_camera_from_tensor -> SLAMNetwork.forward(mode="tracking") -> L1 -> loss.backward() -> torch.optim.Adam.step().
Per phase: host ISSUE time (perf_counter around the call, no synchronisation -- what the Python thread spends before it can issue the
next phase), and the synchronised wall time per iteration.  Run twice: with the cached hipGraphs of fused/track_graph.py (default) and
with NSA_TRACK_GRAPH=0 (the eager autograd.Functions of round 3).
   python tools/profile_dropin_host.py > profiles/r04_dropin_host.txt"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from nicer_slam_amd.fused import track_graph
from nicer_slam_amd.utils.general import get_camera_from_tensor


def run(graph, iters=300):
    dev = torch.device("cuda", 0)
    args = argparse.Namespace(samples=128, engine="auto", precision="fp32", param_grads=True)
    model, conf = bench.make_model(args, dev)          # every parameter requires grad, as the reference builds the model
    track_graph.ENABLED = graph
    K = torch.eye(4, device=dev)
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    K = K[None]
    gen = torch.Generator(device=dev).manual_seed(1)
    batches = [bench.synth_batch(gen, 1024, dev) for _ in range(64)]
    cam = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2], device=dev).requires_grad_(True)
    opt = torch.optim.Adam([cam], lr=0.005)
    ind = torch.zeros(1, dtype=torch.long, device=dev)
    phases = ["get_camera_from_tensor", "model.forward", "L1 loss", "loss.backward()", "optimizer.step + zero_grad"]
    acc = [0.0] * len(phases)

    def it(i, timed):
        uv, gt = batches[i % len(batches)]
        t = [time.perf_counter()]
        pose = get_camera_from_tensor(cam).unsqueeze(0)
        t.append(time.perf_counter())
        out = model({"intrinsics": K, "uv": uv, "pose": pose}, ind, {}, mode="tracking", frame_idx=1)
        t.append(time.perf_counter())
        loss = (out["rgb_values"].reshape(-1, 3) - gt).abs().mean()
        t.append(time.perf_counter())
        loss.backward()
        t.append(time.perf_counter())
        opt.step()
        opt.zero_grad()
        t.append(time.perf_counter())
        if timed:
            for k in range(len(phases)):
                acc[k] += t[k + 1] - t[k]

    with bench.quiet_gc():
        for i in range(30):
            it(i, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(iters):
            it(i, True)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    print(f"== cached hipGraphs {'ON' if graph else 'OFF (NSA_TRACK_GRAPH=0: eager autograd.Functions)'}: {iters} iterations, "
          f"engine {model.last_engine}")
    print(f"   wall per iteration (synchronised at the end): {wall / iters * 1e3:.3f} ms    host issue per iteration: {host / iters * 1e3:.3f} ms"
          f"   -> {'HOST-bound' if host > 0.97 * wall else 'device-bound'}")
    for name, a in zip(phases, acc):
        print(f"   {a / iters * 1e6:8.1f} us  {name}")
    del model
    torch.cuda.empty_cache()


if __name__ == "__main__":
    a = torch.randn(4096, 4096, device="cuda")
    t0 = time.time()
    while time.time() - t0 < 1.0:
        a @ a
    torch.cuda.synchronize()
    run(True)
    run(False)
