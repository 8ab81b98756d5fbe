import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from helpers import load, tt, params_of, oracle_config, draws_of, golden_objective
from test_fused_gpu import _setup
from oracle import render_ref as R
from nicer_slam_amd.fused import render as fr

name = sys.argv[1] if len(sys.argv) > 1 else "full_tracking_poisson"
fx, model, cam, pose, rays_o, rays_d = _setup(name)
model.train(True)
z = tt(fx["out_z_vals"]).cuda()
ro, rd = rays_o.detach().contiguous(), rays_d.detach().contiguous()
b = fr.composite_forward_raw(model, ro, rd, z, "fine", True)
n_ray, S = z.shape
gt = tt(fx["gt_rgb"]).cuda()
g_rgbv = torch.sign(b["rgb_values"] - gt) / (3 * n_ray)
g_o, g_d = fr.composite_backward_raw(model, ro, rd, z, b, "fine", "highfreq", g_rgbv=g_rgbv.contiguous())
# oracle with per-ray grads
cfg, params = oracle_config(fx), params_of(fx)
oc = ro.cpu().clone().requires_grad_(True); dc = rd.cpu().clone().requires_grad_(True)
pts = (oc.unsqueeze(1) + z.cpu().unsqueeze(2) * dc.unsqueeze(1)).reshape(-1, 3)
pts.retain_grad()
sdf, feat, g = R.sdf_outputs(params, cfg, pts, "fine")
dirs_flat = dc.unsqueeze(1).repeat(1, S, 1).reshape(-1, 3)
rgb = R.colour_net(params, cfg, pts, g, dirs_flat, feat, "highfreq").reshape(-1, S, 3)
w = R.volume_weights(z.cpu(), sdf, pts, tt(fx["in_voxels"]), 64)
rgbv = (w.unsqueeze(-1) * rgb).sum(1)
loss = (rgbv - gt.cpu()).abs().mean()
loss.backward()
def md(a, bb): return float((a.detach().cpu() - bb.detach().cpu()).abs().max())
print("fwd diffs: sdf", md(b["sdf"], sdf.reshape(-1)), "grad", md(b["grad"], g), "rgb", md(b["rgb"], rgb.reshape(-1, 3)), "w", md(b["weights"], w), "rgbv", md(b["rgb_values"], rgbv))
print("bwd diffs: g_o", md(g_o, oc.grad), "max", float(oc.grad.abs().max()), " g_d", md(g_d, dc.grad), "max", float(dc.grad.abs().max()))
do = (g_o.cpu() - oc.grad).abs().max(1)[0]; dd = (g_d.cpu() - dc.grad).abs().max(1)[0]
print("worst rays g_o:", torch.topk(do, 3), " g_d:", torch.topk(dd, 3))
r = int(torch.argmax(dd))
print("ray", r, "z", z[r].cpu().numpy())
P = pts.detach().reshape(n_ray, S, 3)[r]
print("pts max abs per sample", P.abs().max(1)[0].numpy())
print("weights", w[r].detach().numpy())
vox = ((P + 1) / 2 * 64)
print("voxel coords frac dist to boundary (min over dims)", (vox - vox.round()).abs().min(1)[0].numpy())

# ---- per-sample backward pieces for the worst ray
import ctypes
from nicer_slam_amd._native import lib, check
from nicer_slam_amd.fused.sampler import grid_desc
imp = model.implicit_network
gc, k1 = grid_desc(imp.coarse.encoding, imp.coarse.divide_factor, 1)
gf, k2 = grid_desc(imp.fine.encoding, imp.fine.divide_factor, 3)
gr, k3 = grid_desc(model.rendering_network.encoding, model.rendering_network.divide_factor, 2)
pc, pf, pr = b["packs"]
ptsd = fr._pts(ro, rd, z)
P = n_ray * S
st = torch.cuda.current_stream().cuda_stream
g_sdf = torch.empty(P, device="cuda"); g_rgb = torch.empty(P, 3, device="cuda"); g_grad = torch.empty(P, 3, device="cuda")
check(lib.nsa_composite_backward(ro.data_ptr(), rd.data_ptr(), z.data_ptr(), b["sdf"].data_ptr(), b["rgb"].data_ptr(), b["grad"].data_ptr(),
      b["vox"].data_ptr(), 64, n_ray, S, g_rgbv.contiguous().data_ptr(), None, None, None, None, g_sdf.data_ptr(), g_rgb.data_ptr(), g_grad.data_ptr(), st))
# oracle intermediate grads
sdf.retain_grad() if sdf.requires_grad else None
oc2 = ro.cpu().clone(); dc2 = rd.cpu().clone()
pts2 = (oc2.unsqueeze(1) + z.cpu().unsqueeze(2) * dc2.unsqueeze(1)).reshape(-1, 3).requires_grad_(True)
sdf2, feat2, g2 = R.sdf_outputs(params, cfg, pts2, "fine")
sdf2.retain_grad(); g2.retain_grad(); feat2.retain_grad()
rgb2 = R.colour_net(params, cfg, pts2, g2, dirs_flat.detach(), feat2, "highfreq").reshape(-1, S, 3)
rgb2.retain_grad()
w2 = R.volume_weights(z.cpu(), sdf2, pts2, tt(fx["in_voxels"]), 64)
((w2.unsqueeze(-1) * rgb2).sum(1) - gt.cpu()).abs().mean().backward()
sl = slice(r * S, (r + 1) * S)
print("g_sdf fused ", g_sdf[sl].cpu().numpy())
print("g_sdf oracle", sdf2.grad.reshape(-1)[sl].numpy())
print("g_rgb diff", md(g_rgb[sl], rgb2.grad.reshape(-1, 3)[sl]))
g_feat = torch.empty(fr.hl_size(P), device="cuda"); g_x = torch.empty(P, 3, device="cuda"); g_dir = torch.empty(P, 3, device="cuda")
check(lib.nsa_colour_backward(ctypes.byref(ptsd), ctypes.byref(gr), pr.data_ptr(), b["grad"].data_ptr(), b["feat"].data_ptr(), b["save"].data_ptr(),
      g_rgb.data_ptr(), 1, g_feat.data_ptr(), g_grad.data_ptr(), g_x.data_ptr(), g_dir.data_ptr(), st))
print("g_grad(after colour) diff ray", md(g_grad[sl], g2.grad[sl]), "max", float(g2.grad[sl].abs().max()))
dense = g_feat[fr.hl_index(P, "cuda")]
print("g_feat diff ray", md(dense[sl], feat2.grad[sl]), "max", float(feat2.grad[sl].abs().max()))
i17 = r * S + S - 1
print("g_sdf diff ray", md(g_sdf[sl], sdf2.grad.reshape(-1)[sl]), " at last:", float(g_sdf[i17]), float(sdf2.grad.reshape(-1)[i17]))
print("cot at last: g_grad", g_grad[i17].cpu().numpy(), "oracle", g2.grad[i17].numpy(), " g_feat max", float(dense[i17].abs().max()))
print("x last", pts2[i17].detach().numpy().tolist(), " fused point via kernel: o", ro[r].cpu().numpy().tolist(), "d", rd[r].cpu().numpy().tolist(), "z", float(z[r, -1]))
print("g_x after colour ", g_x[i17].cpu().numpy())
check(lib.nsa_sdfnet_backward(ctypes.byref(ptsd), ctypes.byref(gc), pc.data_ptr(), g_sdf.data_ptr(), g_feat.data_ptr(), g_grad.data_ptr(), 1, g_x.data_ptr(), st))
print("g_x after coarse ", g_x[i17].cpu().numpy())
check(lib.nsa_sdfnet_backward(ctypes.byref(ptsd), ctypes.byref(gf), pf.data_ptr(), g_sdf.data_ptr(), g_feat.data_ptr(), g_grad.data_ptr(), 1, g_x.data_ptr(), st))
print("g_x after fine   ", g_x[i17].cpu().numpy(), " oracle", pts2.grad[i17].numpy())
dx = (g_x[sl].cpu() - pts2.grad[sl]).abs().max(1)[0]
print("g_x per-sample diff", dx.numpy())
print("g_x oracle max per sample", pts2.grad[sl].abs().max(1)[0].numpy())
