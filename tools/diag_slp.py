"""Diagnostic for the SLP (v_pk_*_f32) irreproducibility of the quad-tiling kernels: which points differ between two
runs of the same launch, by how much, and in which sub-network.  NSA_LIB_TAG=slpnop python tools/diag_slp.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nicer_slam_amd.model.network import SLAMNetwork
from nicer_slam_amd.utils.conf import replica_model_conf
from nicer_slam_amd import inference

torch.manual_seed(0)
model = SLAMNetwork(replica_model_conf(94, 640, 32, use_warp_loss=False), n_images=1,
                    colour_grid=dict(base_resolution=16, desired_resolution=64, log2_hashmap_size=12)).cuda().train()
g = torch.Generator(device="cuda").manual_seed(3)
with torch.no_grad():
    for enc in (model.implicit_network.coarse.encoding, model.implicit_network.fine.encoding):
        enc.embeddings.copy_((torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1) * 0.05)
    for n_, p in model.named_parameters():
        if n_.startswith("implicit_network") and n_.endswith("weight_v"):
            p.add_(0.05 * torch.randn(p.shape, device="cuda", generator=g))
N = 640000
pts = (torch.rand(N, 3, device="cuda", generator=g) * 2 - 1) * 0.9
model.sdf_tile = 32
ref = {s: inference.sdf_values(model, pts, s).clone() for s in ("coarse", "fine")}
model.sdf_tile = 16
print("lib", os.environ.get("NSA_LIB_TAG", "(product)"))
for stage in ("coarse", "fine"):
    runs = [inference.sdf_values(model, pts, stage).clone() for _ in range(6)]
    bad = torch.zeros(N, dtype=torch.bool, device="cuda")
    for r in runs[1:]:
        bad |= r != runs[0]
    idx = bad.nonzero().flatten()
    print(f"stage {stage}: {idx.numel()} points differ between runs")
    if idx.numel() == 0:
        continue
    stack = torch.stack(runs)[:, idx]                              # [6, n]
    spread = (stack.max(0).values - stack.min(0).values)
    err = (stack - ref[stage][idx]).abs()                          # vs the 32-point tiling (fp32-faithful, agrees to ~1e-6)
    print(f"   spread between runs: median {float(spread.median()):.3g}  max {float(spread.max()):.3g};  "
          f"|value - 32-point kernel|: min over runs median {float(err.min(0).values.median()):.3g}, max over runs median {float(err.max(0).values.median()):.3g}")
    n_right = (err < 2e-5).sum(0)
    print("   runs (of 6) in which a differing point has the RIGHT value: histogram", torch.bincount(n_right, minlength=7).tolist())
    j = idx % 16
    tile = idx // 16
    print("   point-in-tile j histogram:", torch.bincount(j, minlength=16).tolist())
    per_tile = torch.bincount(torch.bincount(tile)[torch.bincount(tile) > 0], minlength=17).tolist()
    print("   differing points per affected tile (count of tiles with k bad points, k=0..16):", per_tile)
    wave_in_wg = (tile % 12)
    print("   wave-in-workgroup (tile % 12) histogram:", torch.bincount(wave_in_wg, minlength=12).tolist())
    blk = (tile // 12) % 256
    print("   workgroups (CUs) touched:", int((torch.bincount(blk, minlength=256) > 0).sum()), "of 256")
