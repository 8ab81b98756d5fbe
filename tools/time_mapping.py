"""Time one MAPPING iteration (8192 rays, shipped sizes, parameter gradients + eikonal samples + Adam over all groups)
on the current engines -- context for DESIGN.md; the headline metric is the tracking iteration (bench.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nicer_slam_amd.model.network import SLAMNetwork
from nicer_slam_amd.utils.conf import replica_model_conf
from nicer_slam_amd.utils.general import get_camera_from_tensor


class DS:
    img_res = (680, 1200)


torch.manual_seed(0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ENGINE = sys.argv[2] if len(sys.argv) > 2 else "fused"       # "fused" | "composed"
bs = 8
model = SLAMNetwork(replica_model_conf(64, 640, 32, use_warp_loss=False), dataset=DS(), n_images=2000).cuda().train()
if ENGINE == "fused":
    model.freeze_fine_mlp()          # as the reference's optimizer list (volsdf_train.py:150-173)
model.engine = "auto" if ENGINE == "fused" else "composed"
groups = [{"params": list(model.implicit_network.fine.grid_parameters()), "lr": 0.04},
          {"params": list(model.implicit_network.coarse.grid_parameters()), "lr": 0.04},
          {"params": list(model.rendering_network.grid_parameters()), "lr": 0.01},
          {"params": list(model.rendering_network.mlp_parameters()), "lr": 0.002},
          {"params": list(model.implicit_network.coarse.mlp_parameters()), "lr": 0.002}]
if ENGINE == "fused":
    from nicer_slam_amd.optim import Adam
    opt = Adam(groups, betas=(0.9, 0.99), eps=1e-15)
else:
    opt = torch.optim.Adam(groups, betas=(0.9, 0.99), eps=1e-15)
g = torch.Generator(device="cuda").manual_seed(1)
K = torch.eye(4, device="cuda"); K[0, 0] = K[1, 1] = 600.0; K[0, 2], K[1, 2] = 599.5, 339.5
K = K[None].repeat(bs, 1, 1)
cams = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2], device="cuda").repeat(bs, 1) + 0.01 * torch.randn(bs, 7, device="cuda", generator=g)


def step():
    idx = torch.randint(680 * 1200, (bs, R // bs), device="cuda", generator=g)
    uv = torch.stack([(idx % 1200).float(), (idx // 1200).float()], -1)
    gt = torch.rand(R, 3, device="cuda", generator=g)
    opt.zero_grad()
    out = model({"intrinsics": K, "uv": uv, "pose": get_camera_from_tensor(cams)}, torch.arange(bs, device="cuda"), {},
                mode="mapping", stage="fine", color_stage="highfreq", frame_idx=5)
    loss = (out["rgb_values"].reshape(-1, 3) - gt).abs().mean()
    loss = loss + 0.1 * ((out["grad_theta"].norm(2, dim=1) - 1) ** 2).mean()
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"mapping iteration: {R} rays, engine {model.last_engine}: {dt * 1e3:.1f} ms  ({R / dt:.0f} rays/s)  loss {float(l):.4f}")
print("max memory GB", torch.cuda.max_memory_allocated() / 2 ** 30)

if os.environ.get("NSA_SYNC_DEBUG"):
    # where does the host wait for the device inside one iteration?  (torch's sync debug mode + python stacks), and how long
    # does the host need to ISSUE an iteration (CPU time until the last launch returns, device still running)
    import traceback, warnings
    seen = {}

    def show(message, category, filename, lineno, file=None, line=None):
        st = [f for f in traceback.extract_stack()[:-2] if "nicer_slam_amd" in f.filename or "time_mapping" in f.filename]
        key = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(st[-4:]))
        seen[key] = seen.get(key, 0) + 1

    warnings.showwarning = show
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode(1)
    step()
    torch.cuda.set_sync_debug_mode(0)
    for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
        print(f"sync x{v}: {k}")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host issue time {1e3 * (t1 - t0):.2f} ms, device drained after {1e3 * (t2 - t0):.2f} ms")
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
