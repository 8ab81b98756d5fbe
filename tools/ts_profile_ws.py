"""Where do the cycles of the wave-specialised sampler go?  Profiling build:
   NSA_BUILD_TAG=ts NSA_EXTRA_HIPCC_FLAGS=-DNSA_X_TS python -m nicer_slam_amd.build ;  NSA_LIB_TAG=ts python tools/ts_profile_ws.py
Mean cycles per 32-point tile of a V wave's phases and per GEMM of an M wave's (s_memtime, ~10 % overhead)."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from nicer_slam_amd._native import lib
from nicer_slam_amd.model.network import SLAMNetwork
from nicer_slam_amd.utils.conf import replica_model_conf
from nicer_slam_amd.fused import sampler as fs

V = {0: "ray, z, point, PE, coarse grid gather", 1: "split + publish B (coarse layer 0)", 2: "fine grid gather",
     3: "WAIT for accumulators (4 per tile)", 4: "acc read + softplus + sdf dot (coarse)", 5: "split + publish B (fine layer 0)",
     6: "acc read + softplus (2 hidden layers)", 7: "split + publish B (2 hidden layers)", 8: "last softplus + sdf dot + stores"}
M = {0: "WAIT for B fragments", 1: "bias + B reads + MFMAs", 2: "WAIT for the partner's reads", 3: "accumulator write + flag"}


def main():
    torch.manual_seed(0)
    model = SLAMNetwork(replica_model_conf(94, 640, 32, use_warp_loss=False), n_images=1).cuda().train()
    g = torch.Generator(device="cuda").manual_seed(3)
    R = 1024
    d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda", generator=g), dim=-1) * 0.7
    o = (torch.rand(R, 3, device="cuda", generator=g) - 0.5) * 0.4
    t_rand = torch.rand(R, 640, device="cuda", generator=g)
    model.sdf_tile = 96
    for _ in range(5):
        fs.sampler_sdf(model, o, d, t_rand)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    buf = torch.zeros(n_cu * 16 * 16, dtype=torch.int64, device="cuda")
    lib.nsa_debug_set_ts_ws.argtypes = [ctypes.c_void_p]
    assert lib.nsa_debug_set_ts_ws(buf.data_ptr()) == 0
    fs.sampler_sdf(model, o, d, t_rand)
    torch.cuda.synchronize()
    lib.nsa_debug_set_ts_ws(None)
    t = buf.view(n_cu, 16, 16).double()
    for name, sl, slots in (("M waves (per GEMM)", slice(0, 8), M), ("V waves (per tile)", slice(8, 16), V)):
        w = t[:, sl].reshape(-1, 16)
        cnt = w[:, 14].sum().item()
        life = w[:, 15].mean().item()
        print(f"{name}: {w.shape[0]} waves, mean lifetime {life:.0f} cycles, {cnt / w.shape[0]:.1f} units per wave, "
              f"{w[:, 15].sum().item() / cnt:.0f} cycles per unit")
        for k, what in slots.items():
            v = w[:, k].sum().item()
            print(f"   {v / cnt:8.0f}  {100 * v / w[:, 15].sum().item():5.1f} %   {what}")
    for j in range(8):
        w = t[:, j]
        print(f"   M wave {j}: busy (GEMM + write) {100 * (w[:, 1] + w[:, 3]).sum().item() / w[:, 15].sum().item():.1f} % of its lifetime")


if __name__ == "__main__":
    main()
