#!/usr/bin/env python
"""Where do a wave's cycles go?  Needs the profiling build (NSA_BUILD_TAG=ts NSA_EXTRA_HIPCC_FLAGS=-DNSA_X_TS python -m
nicer_slam_amd.build; run with NSA_LIB_TAG=ts): the quad SDF forward kernel keeps per-wave cycle counts (s_memtime) of its
phases and of every staged-GEMM wait in LDS and writes them out at the end.  Prints the mean over all waves of the fine-network
forward launch at the bench shape.  Development tool."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nicer_slam_amd.fused import sampler as _fs
_fs.FWD_PAIR = False     # the instrumented forward is the single-network kernel (two launches)

SLOTS_BWD = {0: "stage_wait: vmcnt(0)", 1: "stage_wait: barrier", 2: "stage_wait count", 3: "GEMM bodies (LDS reads + split + MFMA)",
             4: "point load + geometry barrier", 5: "PE + grid gather/blend/Jacobian", 6: "forward recompute (hidden layers)",
             7: "reverse pass (recompute)", 8: "cotangent loads + tangent sweep", 9: "feature-cotangent load + reverse sweep",
             10: "first-layer transposed GEMM + d/dx + output", 15: "whole kernel"}
SLOTS = {0: "stage_wait: vmcnt(0)", 1: "stage_wait: barrier", 2: "stage_wait count", 3: "GEMM bodies (LDS reads + split + MFMA)",
         4: "point load + geometry barrier", 5: "PE + grid gather/blend/Jacobian", 6: "hidden layers (incl. their GEMMs and waits)",
         7: "sdf dot + feature GEMM + feature store", 8: "reverse pass (incl. GEMMs and waits)", 9: "grad assembly + output",
         15: "whole kernel"}


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
    n_waves = 131072 // 16 + 64
    buf = torch.zeros(n_waves * 16, dtype=torch.int64, device=dev)
    lib.nsa_debug_set_ts.argtypes = [ctypes.c_void_p]
    for i in range(3):
        tr.step(*batches[i])
    torch.cuda.synchronize()
    assert lib.nsa_debug_set_ts(buf.data_ptr()) == 0
    tr.step(*batches[3])                  # fwd: the fine launch overwrites the coarse one's rows; bwd (quad, fine only) runs later
                                          # in the step and overwrites the forward's -> `--fwd` profiles a forward-only call
    torch.cuda.synchronize()
    lib.nsa_debug_set_ts(None)
    t = buf.view(n_waves, 16)[:131072 // 16].double()
    tot = t[:, 15].mean().item()
    print(f"waves {t.shape[0]}  mean wave lifetime {tot:.0f} cycles   (last instrumented launch of the step: the fine backward)")
    for k, name in SLOTS_BWD.items():
        v = t[:, k].mean().item()
        print(f"  slot {k:2d}  {v:9.0f}  {100 * v / tot:5.1f} %   {name}")
    # the colour backward (32-point tiles, 4-wave workgroups, staged weights)
    if hasattr(lib, "nsa_debug_set_ts_colour"):
        n_c = 131072 // 32
        bufc = torch.zeros((n_c + 64) * 16, dtype=torch.int64, device=dev)
        lib.nsa_debug_set_ts_colour.argtypes = [ctypes.c_void_p]
        assert lib.nsa_debug_set_ts_colour(bufc.data_ptr()) == 0
        tr.step(*batches[5])
        torch.cuda.synchronize()
        lib.nsa_debug_set_ts_colour(None)
        t = bufc.view(-1, 16)[:n_c].double()
        tot = t[:, 15].mean().item()
        print(f"k_colour_bwd: waves {n_c}  mean wave lifetime {tot:.0f} cycles (min {t[:, 15].min().item():.0f}, max {t[:, 15].max().item():.0f})")
        for k, name in {0: "   of which stage_wait: vmcnt(0)", 1: "   of which stage_wait: barrier", 4: "point, dir, saved features -> inputs",
                        5: "MLP recompute (3 staged GEMMs)", 6: "d/d pre-sigmoid, d/d h2 (VALU)", 7: "W1^T GEMM", 8: "W0^T GEMM (2 parts)",
                        9: "feature-cotangent store, PE / direction terms", 10: "Jacobian loads (save area) + d/dx", 11: "half sums + stores",
                        15: "whole kernel"}.items():
            v = t[:, k].mean().item()
            print(f"  slot {k:2d}  {v:9.0f}  {100 * v / tot:5.1f} %   {name}")
    # the sampler (two 32-point tiles per wave)
    n_s = 1024 * 640 // 64
    buf2 = torch.zeros((n_s + 64) * 16, dtype=torch.int64, device=dev)
    lib.nsa_debug_set_ts_sampler.argtypes = [ctypes.c_void_p]
    assert lib.nsa_debug_set_ts_sampler(buf2.data_ptr()) == 0
    tr.step(*batches[4])
    torch.cuda.synchronize()
    lib.nsa_debug_set_ts_sampler(None)
    t = buf2.view(-1, 16)[:n_s].double()
    tot = t[:, 15].mean().item()
    print(f"sampler: waves {n_s}  mean wave lifetime {tot:.0f} cycles")
    for k, name in {4: "ray + stratified z + point (both tiles)", 5: "positional encoding", 6: "coarse grid gather + blend",
                    7: "coarse MLP (W0 GEMM, softplus, sdf dot)", 8: "fine grid gather + blend", 9: "fine MLP (3 GEMMs)",
                    10: "stores", 15: "whole kernel"}.items():
        v = t[:, k].mean().item()
        print(f"  slot {k:2d}  {v:9.0f}  {100 * v / tot:5.1f} %   {name}")
    # the per-ray sampling kernel ran after the sampler in the same step and overwrote the first 4 x 1024 rows
    t = buf2.view(-1, 16)[:4096].double()
    tot = t[:, 15].mean().item()
    print(f"k_sample_rays: waves 4096  mean wave lifetime {tot:.0f} cycles")
    for k, name in {1: "ray / z / sdf loads, voxel gather, density, free energy -> LDS", 2: "block sums + scan of the free energy",
                    3: "weights, pdf", 4: "pdf total", 5: "cdf scan", 6: "inverse-CDF search + extras", 7: "rank sort + store",
                    15: "whole kernel"}.items():
        v = t[:, k].mean().item()
        print(f"  slot {k:2d}  {v:9.0f}  {100 * v / tot:5.1f} %   {name}")


if __name__ == "__main__":
    main()
