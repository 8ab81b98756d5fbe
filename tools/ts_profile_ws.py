"""Where do the cycles of the wave-specialised sampler go?  Profiling build:
   NSA_BUILD_TAG=ts NSA_X_WS=1 NSA_EXTRA_HIPCC_FLAGS=-DNSA_X_TS python -m nicer_slam_amd.build ;  NSA_LIB_TAG=ts python tools/ts_profile_ws.py
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


VS = {0: "ray, z, point", 1: "positional encoding", 2: "coarse grid gather", 3: "fine grid gather + z stores", 6: "ticket + WAIT for a free tile record",
      4: "WAIT for a free coarse first-layer slot", 7: "WAIT for a free fine first-layer slot", 5: "split + publish B (coarse / fine layer 0)"}
ES = {0: "WAIT for B fragments", 1: "bias + B reads + MFMAs", 2: "WAIT (partner's dot half / sdf_c / ring entry free)",
      3: "softplus + dot or split + publish"}


def main():
    sys_form = "--sys" in sys.argv
    torch.manual_seed(0)
    model = SLAMNetwork(replica_model_conf(94, 640, 32, use_warp_loss=False), n_images=1).cuda().train()
    g = torch.Generator(device="cuda").manual_seed(3)
    R = 1024
    d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda", generator=g), dim=-1) * 0.7
    o = (torch.rand(R, 3, device="cuda", generator=g) - 0.5) * 0.4
    t_rand = torch.rand(R, 640, device="cuda", generator=g)
    model.sdf_tile = 97 if sys_form else 96
    for _ in range(5):
        fs.sampler_sdf(model, o, d, t_rand)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    buf = torch.zeros(n_cu * 16 * 16, dtype=torch.int64, device="cuda")
    setter = lib.nsa_debug_set_ts_sys if sys_form else lib.nsa_debug_set_ts_ws
    setter.argtypes = [ctypes.c_void_p]
    assert setter(buf.data_ptr()) == 0
    fs.sampler_sdf(model, o, d, t_rand)
    torch.cuda.synchronize()
    setter(None)
    t = buf.view(n_cu, 16, 16).double()
    if sys_form and "--trace" in sys.argv:
        tr = torch.zeros(96 * 32, dtype=torch.int64, device="cuda")
        lib.nsa_debug_set_trace_sys.argtypes = [ctypes.c_void_p]
        assert lib.nsa_debug_set_trace_sys(tr.data_ptr()) == 0
        fs.sampler_sdf(model, o, d, t_rand)
        torch.cuda.synchronize()
        lib.nsa_debug_set_trace_sys(None)
        e = tr.view(96, 32).cpu()
        t0 = int(e[:, 0][e[:, 0] > 0].min())
        print("per tile q of workgroup 0, cycles since the first ticket: ticket | slotC got | B1 ready | slotF got | B2 ready || "
              "C0m0 in/out | C0m1 in/out | F0m0 | F0m1 | F1m0 | F1m1 | F2m0 | F2m1")
        for q in range(40, 64):
            r = [int(x) - t0 if int(x) else -1 for x in e[q].tolist()]
            print(f"q{q:3d}: {r[0]:7d} {r[1]:7d} {r[2]:7d} {r[3]:7d} {r[4]:7d} || " +
                  " | ".join(f"{r[8 + st * 4 + mt * 2]:7d}/{r[8 + st * 4 + mt * 2 + 1]:7d}" for st in range(4) for mt in range(2)))
        return
    if sys_form:
        names = ["C0 mt0", "F0 mt0", "F1 mt0", "F2 mt0", "F1 mt1", "F2 mt1", "C0 mt1", "F0 mt1"]     # wave w: stage / tile as in k_sampler_sys
        for j in range(8):
            w = t[:, j]
            cnt = w[:, 14].sum().item()
            print(f"engine wave {j} ({names[j]}): {w[:, 15].sum().item() / cnt:.0f} cycles per tile: " +
                  ", ".join(f"{ES[k]} {w[:, k].sum().item() / cnt:.0f}" for k in ES))
        w = t[:, 8:16].reshape(-1, 16)
        cnt = w[:, 14].sum().item()
        print(f"V waves: {w[:, 15].sum().item() / cnt:.0f} cycles per tile")
        for k, what in VS.items():
            print(f"   {w[:, k].sum().item() / cnt:8.0f}  {100 * w[:, k].sum().item() / w[:, 15].sum().item():5.1f} %   {what}")
        return
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
