"""Self-consistency demo of the tracking hot path: render target colours from the model at a known camera, perturb the camera, run
KernelTracker iterations with the reference's optimizer settings (Adam lr 5e-3 here, StepLR(50, 0.95), arg-min-loss candidate) and
report how the pose error falls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse
import torch
import bench
from nicer_slam_amd.tracking import KernelTracker
from nicer_slam_amd.utils.general import get_camera_from_tensor



def run(iters=200, verbose=True):
    """-> (start errors, final errors, candidate errors) as (rotation, translation) pairs"""
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    args = argparse.Namespace(samples=128, engine="auto", precision="fp32", param_grads=False)
    model, conf = bench.make_model(args, dev)
    # give the random-initialised colour network some structure to track against
    with torch.no_grad():
        model.rendering_network.encoding.embeddings.uniform_(-0.5, 0.5)
    K = torch.eye(4, device=dev)
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    cam_true = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2], device=dev)
    R = 1024
    g = torch.Generator(device=dev).manual_seed(3)
    H, W = bench.DS.img_res

    def batch():
        idx = torch.randint(H * W, (1, R), device=dev, generator=g)
        return torch.stack([(idx % W).float(), (idx // W).float()], -1)

    def render(cam, uv):
        model.eval()
        with torch.no_grad():
            out = model({"intrinsics": K[None], "uv": uv, "pose": get_camera_from_tensor(cam).unsqueeze(0)},
                        torch.zeros(1, dtype=torch.long, device=dev), {}, mode="tracking_vis", frame_idx=1)
        model.train()
        return out["rgb_values"].reshape(-1, 3)

    cam0 = cam_true + torch.tensor([0.0, 0.01, -0.008, 0.006, 0.02, -0.015, 0.01], device=dev)
    kt = KernelTracker(model, K, R, cam0, lr=0.002, lr_step=50, lr_gamma=0.95, use_graph=True)
    err = lambda c: (float((c[:4] / c[:4].norm() - cam_true[:4]).norm()), float((c[4:] - cam_true[4:]).norm()))
    start = err(cam0)
    if verbose:
        print("start: rot err %.4f  trans err %.4f" % start)
    for it in range(iters):
        uv = batch()
        loss = kt.step(uv, render(cam_true, uv))
        if verbose and it % 25 == 24:
            print("iter %3d  loss %.5f  rot err %.4f  trans err %.4f" % ((it + 1, float(loss)) + err(kt.cam)))
    final, cand = err(kt.cam), err(kt.candidate)
    if verbose:
        print("candidate (arg-min loss): rot err %.4f  trans err %.4f  min loss %.5f" % (cand + (float(kt.min_loss),)))
    return start, final, cand


if __name__ == "__main__":
    run()
