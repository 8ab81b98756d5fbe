#!/usr/bin/env python
"""Diagnostic (development): points where the quad and the 32-point sampler kernels disagree, against the CPU oracle."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nicer_slam_amd.model.network import SLAMNetwork
from nicer_slam_amd.utils.conf import replica_model_conf
from nicer_slam_amd.fused import sampler as fs
from oracle import render_ref as R

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
Rn = 1000
d = torch.nn.functional.normalize(torch.randn(Rn, 3, device="cuda", generator=g), dim=-1) * 0.7
o = (torch.rand(Rn, 3, device="cuda", generator=g) - 0.5) * 0.4
t_rand = torch.rand(Rn, 640, device="cuda", generator=g)
res = {}
for tile in (16, 32):
    model.sdf_tile = tile
    res[tile] = fs.sampler_sdf(model, o, d, t_rand)
z = res[16][0]
diff = (res[16][1] - res[32][1]).abs()
bad = (diff > 1e-5).nonzero()
print("n bad", bad.shape[0], " max diff", float(diff.max()))
pts = (o.unsqueeze(1) + z.unsqueeze(2) * d.unsqueeze(1))
sel = bad[:12]
x = pts[sel[:, 0], sel[:, 1]].cpu()
mk = R.make_grid_spec
cfg = R.RenderConfig(coarse=R.SdfNetSpec(mk(4, 8, 32, 32, 19), 2), fine=R.SdfNetSpec(mk(8, 4, 32, 128, 19), 4),
                     colour_grid=mk(16, 2, 16, 64, 12), n_samples=94, n_samples_eval=640, n_samples_extra=32)
params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
with torch.no_grad():
    ref = R.sdf_vals(params, cfg, x.clone()).reshape(-1)
    ref_c = R.sdf_vals(params, cfg, x.clone(), "coarse").reshape(-1)
for i in range(sel.shape[0]):
    r, s = int(sel[i, 0]), int(sel[i, 1])
    print(f"ray {r:4d} sample {s:3d} (pid%16 {(r * 640 + s) % 16:2d}, pid%32 {(r * 640 + s) % 32:2d}) x {[round(v, 6) for v in x[i].tolist()]} "
          f"t16 {float(res[16][1][r, s]):+.6f} t32 {float(res[32][1][r, s]):+.6f} oracle {float(ref[i]):+.6f} (coarse part {float(ref_c[i]):+.6f})")
print("bad per ray histogram (first 10 rays with bad):", torch.unique(bad[:, 0], return_counts=True)[0][:10].tolist(),
      torch.unique(bad[:, 0], return_counts=True)[1][:10].tolist())
print("max |x| of bad points", float(pts[bad[:, 0], bad[:, 1]].abs().max()), " min", float(pts[bad[:, 0], bad[:, 1]].abs().amax(1).min()))

# which grid level carries the disagreement: keep one level's rows, zero the rest
for which in ("coarse", "fine"):
    enc = getattr(model.implicit_network, which).encoding
    full = enc.embeddings.data.clone()
    offs = enc.offsets.cpu().tolist()
    for l in range(enc.num_levels):
        enc.embeddings.data.zero_()
        enc.embeddings.data[offs[l]:offs[l + 1]] = full[offs[l]:offs[l + 1]]
        out = {}
        for tile in (16, 32):
            model.sdf_tile = tile
            out[tile] = fs.sampler_sdf(model, o, d, t_rand)[1]
        dd = (out[16] - out[32]).abs()
        print(f"{which} level {l} rows {offs[l + 1] - offs[l]:7d}: n bad {int((dd > 1e-5).sum()):4d} max diff {float(dd.max()):.2e}")
    enc.embeddings.data.copy_(full)

# the same points through the other 32-point kernels: k_sdf_points (streams weights like the sampler) and k_sdfnet_fwd (staged)
import ctypes
from nicer_slam_amd import inference
from nicer_slam_amd.fused import render as fr
from nicer_slam_amd._native import lib, check, PointsDesc
allpts = pts.reshape(-1, 3).contiguous()
ref16 = res[16][1].reshape(-1)
for tile in (16, 32):
    model.sdf_tile = tile
    v = inference.sdf_values(model, allpts, "fine")
    dd = (v - ref16).abs()
    print(f"k_sdf_points tile {tile}: n bad vs quad sampler {int((dd > 1e-5).sum())} max {float(dd.max()):.2e}")
    N = allpts.shape[0]
    pd = PointsDesc(None, None, None, allpts.data_ptr(), N, 0, None)
    sdf = torch.empty(N, device="cuda"); grad = torch.empty(N, 3, device="cuda"); feat = torch.empty(fr.hl_size(N), device="cuda")
    for which, acc in (("coarse", 0), ("fine", 1)):
        gd, keep = fs.sdf_grid_desc(model, which)
        check(lib.nsa_sdfnet_forward(ctypes.byref(pd), ctypes.byref(gd), fs.packed_sdf(model, which).data_ptr(), acc, sdf.data_ptr(),
                                     grad.data_ptr(), feat.data_ptr(), torch.cuda.current_stream().cuda_stream))
    dd = (sdf - ref16).abs()
    print(f"k_sdfnet_fwd tile {tile}: n bad vs quad sampler {int((dd > 1e-5).sum())} max {float(dd.max()):.2e}")

# determinism: the same kernel twice on the same inputs, bitwise
for tile in (16, 32):
    model.sdf_tile = tile
    a1 = fs.sampler_sdf(model, o, d, t_rand)[1].clone()
    a2 = fs.sampler_sdf(model, o, d, t_rand)[1].clone()
    a3 = fs.sampler_sdf(model, o, d, t_rand)[1].clone()
    print(f"determinism tile {tile}: run1 vs run2 differ at {int((a1 != a2).sum())} points, run1 vs run3 {int((a1 != a3).sum())}; "
          f"max {float((a1 - a2).abs().max()):.2e}")
