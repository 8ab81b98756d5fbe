"""Time SLAMLoss forward + backward at the mapping batch shape (8192 rays over 8 keyframes, 98 samples, 180 224 eikonal points):
fused HIP loss kernels (nsa_slam_loss) vs the torch restatement on the same device."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from nicer_slam_amd.model.loss import SLAMLoss
from test_loss_gpu import _random_case

out, gt = _random_case(8, 1024, 98, 22 * 8192, seed=1)
for engine in ("auto", "torch"):
    crit = SLAMLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, smooth_weight=0.005, depth_weight=0.1, normal_l1_weight=0.05,
                    normal_cos_weight=0.05)
    crit.engine = engine

    def step():
        for v in out.values():
            v.grad = None
        crit(out, gt, frame_idx=7, stage="fine")["loss"].backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"SLAMLoss fwd+bwd, engine {engine:5s}: host issue {(t1 - t0) / 20 * 1e3:.3f} ms, device done {(t2 - t0) / 20 * 1e3:.3f} ms per call")
