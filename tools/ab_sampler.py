"""A/B timing of the sampler's SDF pass (nsa_sampler_sdf) at the bench shape for several tile codes, on ONE box:
   python tools/ab_sampler.py [--rays=R] [tiles ...]      (default 1024 rays, tiles 64 32; 96 / 97 need the experiment build:
   NSA_BUILD_TAG=ws NSA_X_WS=1 python -m nicer_slam_amd.build ; NSA_LIB_TAG=ws python tools/ab_sampler.py 64 96 97)
HIP events around back-to-back launches (the kernel runs 100+ us: launch overhead is hidden), GEMM clock pre-warm, three rounds,
the variants interleaved so that clock drift hits them alike.  Also checks that every variant returns the same bits as tile 32."""
import sys
import time

import torch

sys.path.insert(0, ".")
from nicer_slam_amd.model.network import SLAMNetwork
from nicer_slam_amd.utils.conf import replica_model_conf
from nicer_slam_amd.fused import sampler as fs


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--rays=")]
    rays = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--rays=")]
    tiles = [int(t) for t in args] or [64, 32]
    torch.manual_seed(0)
    model = SLAMNetwork(replica_model_conf(94, 640, 32, use_warp_loss=False), n_images=1).cuda().train()
    g = torch.Generator(device="cuda").manual_seed(3)
    with torch.no_grad():
        for enc in (model.implicit_network.coarse.encoding, model.implicit_network.fine.encoding):
            enc.embeddings.copy_((torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1) * 0.05)
        for n_, p in model.named_parameters():
            if n_.startswith("implicit_network") and n_.endswith("weight_v"):
                p.add_(0.05 * torch.randn(p.shape, device="cuda", generator=g))
    R = rays[0] if rays else 1024           # (--rays=8192: the mapping batch; `model.sdf_tile` overrides the by-size choice)
    d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda", generator=g), dim=-1) * 0.7
    o = (torch.rand(R, 3, device="cuda", generator=g) - 0.5) * 0.4
    t_rand = torch.rand(R, 640, device="cuda", generator=g)
    model.sdf_tile = 32
    ref = fs.sampler_sdf(model, o, d, t_rand)
    for t in tiles:
        model.sdf_tile = t
        out = fs.sampler_sdf(model, o, d, t_rand)
        torch.cuda.synchronize()
        bad = [int((a != b).sum()) for a, b in zip(out, ref)]
        print(f"tile {t}: differing elements vs tile 32 (z, sdf, far) = {bad}", flush=True)
    a = torch.randn(4096, 4096, device="cuda")
    t0 = time.time()
    while time.time() - t0 < 1.0:
        a @ a
    torch.cuda.synchronize()
    N = 200 if R <= 1024 else 40
    for rnd in range(3):
        line = []
        for t in tiles:
            model.sdf_tile = t
            for _ in range(20):
                fs.sampler_sdf(model, o, d, t_rand)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(N):
                fs.sampler_sdf(model, o, d, t_rand)
            e1.record()
            torch.cuda.synchronize()
            line.append(f"tile {t}: {e0.elapsed_time(e1) / N * 1e3:7.1f} us")
        print(f"round {rnd}:  " + "   ".join(line), flush=True)


if __name__ == "__main__":
    main()
