#!/usr/bin/env python
"""Diagnostic (development): emission rows H0 / TIN of the coarse MAP backward, quad vs 32-point tiling, same points."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nicer_slam_amd.model.network import SLAMNetwork
from nicer_slam_amd.utils.conf import replica_model_conf
from nicer_slam_amd.fused import sampler as fs, mapping as fm, render as fr
from nicer_slam_amd._native import lib, check, PointsDesc

torch.manual_seed(0)
model = SLAMNetwork(replica_model_conf(use_warp_loss=False), n_images=1,
                    colour_grid=dict(base_resolution=16, desired_resolution=64, log2_hashmap_size=12)).cuda().train()
g = torch.Generator(device="cuda").manual_seed(3)
RAYS = len(sys.argv) > 1 and sys.argv[1] == "rays"
R_, S_ = 232, 57
N = R_ * S_ if RAYS else 5104
pts = ((torch.rand(N, 3, device="cuda", generator=g) * 2 - 1) * 0.9).contiguous()
ro = (torch.rand(R_, 3, device="cuda", generator=g) - 0.5).contiguous() * 0.3
rd = torch.nn.functional.normalize(torch.randn(R_, 3, device="cuda", generator=g), dim=-1).contiguous() * 0.6
zv = torch.sort(torch.rand(R_, S_, device="cuda", generator=g) * 1.2, dim=1).values.contiguous()
order = fr.morton_order(PointsDesc(ro.data_ptr(), rd.data_ptr(), zv.data_ptr(), None, N, S_, None), N, "cuda") if RAYS else None
gg = torch.randn(N, 3, device="cuda", generator=g).contiguous()
gs = torch.randn(N, device="cuda", generator=g).contiguous()
gf = torch.randn(fr.hl_size(N), device="cuda", generator=g).contiguous()
feats = {}
for tile in (16, 32):
    model.sdf_tile = tile
    gd, keep = fs.sdf_grid_desc(model, "coarse")
    m = fm.se_rows(1, tile)
    emit = fm.new_emit(m["ROWS"], N, "cuda")
    gx = torch.empty(N, 3, device="cuda")
    pd = PointsDesc(ro.data_ptr(), rd.data_ptr(), zv.data_ptr(), None, N, S_, order.data_ptr()) if RAYS else PointsDesc(None, None, None, pts.data_ptr(), N, 0, None)
    check(lib.nsa_sdfnet_backward_params(ctypes.byref(pd), ctypes.byref(gd), fs.packed_sdf(model, "coarse").data_ptr(), gs.data_ptr(),
                                         gf.data_ptr(), gg.data_ptr(), 0, gx.data_ptr(), None, emit.data_ptr(), emit.shape[1],
                                         torch.cuda.current_stream().cuda_stream))
    rows = fm._sdf_rows(4, 8, tile).cuda()
    feats[tile] = {"H0": emit[m["H0"]:m["H0"] + m["IN"]][rows][:, :N].clone(), "TIN": emit[m["TIN"]:m["TIN"] + m["IN"]][rows][:, :N].clone(),
                   "AB1": emit[m["AB1"]:m["AB1"] + 64, :N].clone(), "DA1": emit[m["DA1"]:m["DA1"] + 64, :N].clone(),
                   "H1": emit[m["H1"]:m["H1"] + 64, :N].clone(), "TH1": emit[m["TH1"]:m["TH1"] + 64, :N].clone(), "gx": gx.clone()}
for k in feats[16]:
    a, b = feats[16][k], feats[32][k]
    d = (a - b).abs()
    rel = d.amax(1) / (b.abs().amax(1) + 1e-30) if d.dim() == 2 and k != "gx" else d.max() / b.abs().max()
    if k in ("H0", "TIN"):
        bad = (rel > 1e-4).nonzero().flatten().tolist()
        print(k, "max rel err per feature row > 1e-4 at reference features:", bad, [f"{float(rel[i]):.2e}" for i in bad])
    else:
        print(k, "max rel err", float(rel.max()))
