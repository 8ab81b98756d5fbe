import sys, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import load, tt, draws_of
from test_model_cpu import build_model
from nicer_slam_amd.fused import mapping
fx = load("full_mapping")
model = build_model(fx).cuda().freeze_fine_mlp()
model.train(True)
torch.manual_seed(0)
pts = (torch.rand(20000, 3, device="cuda") * 2 - 1)
pts[:100] *= 1.02
g_f = mapping.sdf_gradient(model, pts, "fine")
g_c = model.implicit_network.gradient(pts, stage="fine")
d = (g_f - g_c).abs().max(1).values
bad = (d > 1e-4).nonzero().flatten()
print("n bad", bad.numel(), "max", float(d.max()))
for i in bad[:10].tolist():
    print(i, pts[i].tolist(), g_f[i].tolist(), g_c[i].tolist())
for st in ("coarse",):
    g_f = mapping.sdf_gradient(model, pts, st); g_c = model.implicit_network.gradient(pts, stage=st)
    d = (g_f - g_c).abs().max(1).values
    print(st, "n bad", int((d > 1e-4).sum()), float(d.max()))
