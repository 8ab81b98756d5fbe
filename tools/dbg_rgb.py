import sys, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import load, tt
import test_mapping_gpu as T
fx = load("full_mapping_coarse_base")
model, cam, out = T._run(fx, "fused")
d = (out["rgb"].detach().cpu() - tt(fx["out_rgb"])).abs().amax(-1)
bad = (d > 1e-4).nonzero()
print("bad", bad.tolist(), "shape", tuple(d.shape))
o, dr, z = tt(fx["out_cam_loc"]), tt(fx["out_ray_dirs"]), tt(fx["out_z_vals"])
print(o.shape, dr.shape, z.shape)
for r, s in bad.tolist():
    oo = o.reshape(-1, 3)[r if o.reshape(-1,3).shape[0] > 1 else 0]; dd = dr.reshape(-1, 3)[r]
    print(r, s, (oo + z[r, s] * dd).tolist())
for k in ("rgb_values", "depth_values", "normal_map", "grad_theta"):
    print(k, float((out[k].detach().cpu() - tt(fx["out_" + k])).abs().max()))
