#!/usr/bin/env python
"""Build-container check (needs /root/reference): is the oracle a faithful TIMING proxy for the reference's CPU path?
Times BASELINE configs[0] -- 256 rays x 64 samples, E = 640, shipped SDF grid shapes (the hard-coded 1 GiB colour grid is
replaced by a 2^19-row one on both sides) -- forward + backward to the pose gradient: the reference's own
SLAMNetwork (imported through tests/golden/ref_shims.py, native seam = the C oracle) vs oracle/render_ref.py on the same
parameters, same draws, 8 torch threads.  Prints both times; the number quoted in DESIGN.md section 6 comes from here.

    python oracle/time_vs_reference.py
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import make_golden as G           # noqa: E402  (installs the reference import shims)
from helpers import params_of, oracle_config   # noqa: E402
from oracle import render_ref as R            # noqa: E402
from oracle import hashenc                     # noqa: E402

torch.set_num_threads(8)
hashenc.set_threads(8) if hasattr(hashenc, "set_threads") else None
R_RAYS, N_SAMPLES, E, N_EXTRA = 256, 30, 640, 32          # S = 30 + 2 + 32 = 64
coarse, fine, colour = (32, 32, 19, 4, 8), (32, 128, 19, 8, 4), (16, 2048, 19)
model, conf = G.build_model(1, coarse, fine, colour, N_SAMPLES, E, N_EXTRA, (0.02, 0.02, 0.3))
model.train()
uv, cam0, K = G.synth_inputs(2, 1, R_RAYS)
gt = torch.rand(R_RAYS, 3)


def ref_step():
    cam = cam0.clone().requires_grad_(True)
    pose = G.ref_general.get_camera_from_tensor(cam)
    out = model({"intrinsics": K, "uv": uv, "pose": pose}, torch.arange(1), {}, mode="tracking", frame_idx=1)
    loss = (out["rgb_values"].reshape(-1, 3) - gt).abs().mean()
    loss.backward()
    model.zero_grad(set_to_none=True)
    return float(loss)


def median_time(fn, n=3, warm=1):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


t_ref = median_time(ref_step)
# the oracle on the same parameters (detached: data path only, like a tracking iteration that discards parameter grads)
fx = {"param_" + n: p.detach().numpy() for n, p in model.named_parameters()}
fx.update(meta_coarse_grid=list(coarse), meta_fine_grid=list(fine), meta_colour_grid=list(colour),
          meta_samples=[N_SAMPLES, E, N_EXTRA])
try:
    cfg, params = oracle_config(fx), params_of(fx)
except Exception as e:                                     # helpers expect a loaded fixture: fall back to its loader's format
    raise SystemExit(f"could not build the oracle config from the live model: {e}")
vox = torch.zeros(64, 64, 64)
S = N_SAMPLES + 2 + N_EXTRA
draws = {"t_rand": torch.rand(R_RAYS, E), "extra_idx": torch.randperm(E)[:N_EXTRA], "eik_idx": torch.randint(S, (R_RAYS,))}


def oracle_step():
    cam = cam0.clone().requires_grad_(True)
    out = R.render(params, cfg, uv, R.camera_from_tensor(cam), K, vox, draws, mode="tracking", stage="fine",
                   color_stage="highfreq", training=True)
    loss = (out["rgb_values"].reshape(-1, 3) - gt).abs().mean()
    loss.backward()
    return float(loss)


t_or = median_time(oracle_step)
print(f"reference SLAMNetwork (CPU, 8 threads): {t_ref * 1e3:8.1f} ms / iteration  ({R_RAYS / t_ref:7.1f} rays/s)")
print(f"oracle/render_ref.py   (CPU, 8 threads): {t_or * 1e3:8.1f} ms / iteration  ({R_RAYS / t_or:7.1f} rays/s)")
print(f"ratio oracle / reference = {t_or / t_ref:.2f}")
