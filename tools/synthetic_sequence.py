#!/usr/bin/env python
"""Row g: sequence-level ("matched ATE") evidence without a dataset -- a synthetic multi-frame tracking run.

north_star asks for the speed-up "at matched ATE on Replica room0" and BASELINE configs[3] is an end-to-end ATE-parity config; neither
the dataset nor `pretrain.pth` exists on any box of this build.  What CAN be checked is that the fused engine, driven through the
reference's tracking protocol, recovers a camera TRAJECTORY as well as the composed engine (torch autograd around the stand-alone
hash operator = the reference's op structure) and the CPU oracle do:

  * a "teacher" SLAMNetwork with structured tables (band-limited random colour / SDF grids, perturbed geometric initialisation)
    renders N frames along the first N poses of the reference's ground-truth trajectory of Replica room0
    (tests/golden/replica_room0_traj64.txt = gt_trajs/gt_replica_room0.txt[:64], recentred and scaled into the cube);
  * the map is known (the teacher itself), the cameras are not: every frame >= 1 is tracked exactly as
    code/training/volsdf_train.py:373-446 does -- constant-speed initialisation from the two previous ESTIMATES, Adam(lr 0.005) +
    StepLR(50, 0.95) on the 7-vector, `iters` iterations of `pixels` random pixels through SLAMNetwork.forward(mode="tracking") and the
    rgb-L1 tracking objective, arg-min-loss candidate as the frame's pose;
  * ATE RMSE after rigid alignment as code/evaluation/eval_cam.py:43-105 (Horn / Kabsch, no scale) computes it.

Two experiments:
  (A) free-running: fused (graph-cached forward, the engine's own Philox draws) and composed (torch RNG) -- independent random
      pixels and sampler draws, so the trajectories agree statistically: ATE(fused) within 5 % of ATE(composed) (+ an absolute floor);
  (B) shared draws: fused, composed and -- CPU, first frames, reduced pixel count -- the oracle consume the SAME pixels and sampler
      draws in every iteration, so that per-frame pose differences measure arithmetic only.

    python tools/synthetic_sequence.py [--frames 50 --iters 100 --pixels 1024] [--oracle-frames 3] > profiles/r05_sequence_ate.json
`tests/test_sequence_gpu.py` runs a short version of both."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRAJ = os.path.join(ROOT, "tests", "golden", "replica_room0_traj64.txt")
SCALE = 0.25          # metres -> scene units: Replica room0 (~7 m) inside the cube [-1, 1]^3
# The shipped conf families (nicer_slam_amd/utils/conf.py): camera path, scene scale and default (reduced) frame size of the stand-in
# sequence.  "7scenes" = BASELINE configs[3]'s family: gt_trajs/gt_7scenes_office.txt[:64], 480 x 640 Kinect frames (here halved, as the
# Replica frames are), coarse sphere radius 1.0, fine SDF MLP not geometrically initialised, smooth_weight 0.05.
FAMILIES = {
    "replica": dict(traj=TRAJ, scale=SCALE, size=(340, 600)),
    "7scenes": dict(traj=os.path.join(ROOT, "tests", "golden", "scenes7_office_traj64.txt"), scale=0.4, size=(240, 320)),
    # BASELINE configs[4]'s family: gt_trajs/gt_azure_2.txt[:64] (a hand-held outdoor capture, 1.2 cm per frame), 720 x 1280 frames halved,
    # the 7-Scenes model subtree, assign_scale 15; that config renders 192 samples per ray (--n-samples 158) with a bf16 colour MLP
    "azure": dict(traj=os.path.join(ROOT, "tests", "golden", "azure_2_traj64.txt"), scale=0.2, size=(360, 640)),
}


class _DS:
    def __init__(self, H, W):
        self.img_res = (H, W)


# ------------------------------------------------------------------------------------------------------------ scene and trajectory
def load_trajectory(n, scale=None, path=None, family="replica", stride=1):
    """-> c2w [n,4,4] float32: TUM rows (stamp tx ty tz qx qy qz qw), translations recentred on their mean and scaled.
    stride: every stride-th pose of the fixture (the 7-Scenes office camera starts almost at rest: 0.0009 scene units per frame over its
    first poses, a quarter of Replica room0's -- short test runs take every 4th pose so that there is a motion to recover)."""
    scale = FAMILIES[family]["scale"] if scale is None else scale
    path = FAMILIES[family]["traj"] if path is None else path
    rows = np.loadtxt(path)[::stride][:n]
    assert rows.shape[0] == n, f"the fixture holds {rows.shape[0]} poses"
    t = (rows[:, 1:4] - rows[:, 1:4].mean(0)) * scale
    x, y, z, w = rows[:, 4], rows[:, 5], rows[:, 6], rows[:, 7]
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                  np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                  np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1)
    c2w = np.tile(np.eye(4), (n, 1, 1))
    c2w[:, :3, :3], c2w[:, :3, 3] = R, t
    return torch.from_numpy(c2w.astype(np.float32))


def build_teacher(H, W, n_samples=64, colour_grid=None, seed=5, device="cpu", family="replica"):
    """SLAMNetwork at the shipped sizes (colour_grid: optional smaller colour table for quick runs) with a scene worth tracking:
    band-limited random colour features (amplitude falling with the level), a colour MLP with trained-like gains, small SDF-grid
    features and perturbed first-layer directions (the geometric initialisation alone is a featureless sphere that ignores its encodings)."""
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import model_conf
    torch.manual_seed(seed)
    kw = {} if colour_grid is None else {"colour_grid": colour_grid}
    model = SLAMNetwork(model_conf(family, n_samples, 640, 32, use_warp_loss=False), dataset=_DS(H, W), n_images=64, **kw)
    model.conf_family = family
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for enc, a0, decay in ((model.rendering_network.encoding, 0.8, 0.8), (model.implicit_network.coarse.encoding, 0.03, 1.0),
                               (model.implicit_network.fine.encoding, 0.02, 0.8)):
            offs = [int(o) for o in enc.offsets]
            for l in range(enc.num_levels):
                rows = offs[l + 1] - offs[l]
                enc.embeddings[offs[l]:offs[l + 1]] = (torch.rand(rows, enc.level_dim, generator=g) * 2 - 1) * a0 * decay ** l
        for name, p in model.named_parameters():
            if name.startswith("implicit_network") and name.endswith("weight_v"):
                p.add_(0.02 * torch.randn(p.shape, generator=g))
        # a colour MLP at its default initialisation maps everything to grey (image std 0.005): scale its weight-norm gains as a
        # trained network's would be, so that the rendered frames carry texture (std over an image ~0.27)
        for l, gain in enumerate((5.0, 3.0, 3.0)):
            getattr(model.rendering_network, f"lin{l}").weight_g.mul_(gain)
        if family != "replica":
            # 7-Scenes / Azure confs: `fine.geometric_init = false` -- the fine SDF MLP is built at nn.Linear's default initialisation and
            # then OVERWRITTEN from pretrain.pth (volsdf_train.py:139-147), which does not exist here.  A default-initialised decoder adds a
            # rough +-0.08 field to the coarse sphere; a pretrained residual decoder's output is small: scale the sdf row of its last layer.
            model.implicit_network.fine.lin3.weight_g[0].mul_(0.25)
            model.implicit_network.fine.lin3.bias[0].mul_(0.25)
            # `coarse.bias = 1.0` is the INITIAL radius; the sphere of radius 1 touches the cube's faces, so a scene left there is seen
            # almost only through the far sample of each ray -- the one ON the cube face, whose in-range test hangs on the last ulp of
            # o + z d: frames that no two correct renderers agree on.  The teacher is a TRAINED map: its walls sit at ~0.7,
            # inside the cube as a scaled real scene's do; the student (run_slam) starts from the family's 1.0 and has to pull them in.
            model.implicit_network.coarse.lin1.bias[0].sub_(0.3)
    for p in model.parameters():
        p.requires_grad_(False)
    return model.to(device)


def intrinsics(H, W, device="cpu", family="replica"):
    """the family's shipped camera (Replica: 600, 600, 599.5, 339.5 at 680 x 1200; 7-Scenes: 585, 585, 320, 240 at 480 x 640) scaled to H x W"""
    from nicer_slam_amd.utils.conf import run_conf
    rc = run_conf(family)
    s = H / float(rc["img_res"][0])
    K = torch.eye(4)
    K[0, 0], K[1, 1] = rc["intrinsics"][0] * s, rc["intrinsics"][1] * s
    K[0, 2], K[1, 2] = (W - 1) / 2.0, (H - 1) / 2.0
    return K.to(device)


def all_pixels(H, W, device):
    idx = torch.arange(H * W, device=device)
    return torch.stack([(idx % W).float(), (idx // W).float()], -1)


@torch.no_grad()
def render_frames(model, poses, K, H, W):
    """GT colour images [n, H*W, 3] from the teacher in eval mode (deterministic sampler), fused engine, device tensors."""
    from nicer_slam_amd.inference import render_image
    dev = model.voxels.device
    uv = all_pixels(H, W, dev).unsqueeze(0)
    was = model.training
    model.eval()
    out = []
    for c2w in poses:
        res = render_image(model, {"intrinsics": K[None], "uv": uv, "pose": c2w.to(dev)[None]}, n_pixels=65536)
        out.append(res["rgb_values"].reshape(H * W, 3).clone())
    model.train(was)
    return torch.stack(out)


# ------------------------------------------------------------------------------------------------------------------------- ATE
def align_rigid(model, data):
    """Horn's closed form as code/evaluation/eval_cam.py:43-76: rot, trans minimising |rot model + trans - data| (3 x n arrays)."""
    mc, dc = model - model.mean(1, keepdims=True), data - data.mean(1, keepdims=True)
    Wm = mc @ dc.T                                        # sum of outer(model_i, data_i)
    U, _d, Vh = np.linalg.svd(Wm.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1
    rot = U @ S @ Vh
    trans = data.mean(1, keepdims=True) - rot @ model.mean(1, keepdims=True)
    return rot, trans


def ate_rmse(gt_c2w, est_c2w):
    """ATE RMSE of the camera centres after rigid alignment of the estimate onto the ground truth (eval_cam.py:107-200)."""
    gt = np.asarray(gt_c2w, dtype=np.float64)[:, :3, 3].T
    est = np.asarray(est_c2w, dtype=np.float64)[:, :3, 3].T
    rot, trans = align_rigid(est, gt)
    err = np.linalg.norm(rot @ est + trans - gt, axis=0)
    return float(np.sqrt((err ** 2).mean()))


def rot_err_deg(a, b):
    """angle of a^T b in degrees ([..,3,3])"""
    c = (np.einsum("...ij,...ij->...", np.asarray(a, np.float64), np.asarray(b, np.float64)) - 1) / 2
    return np.degrees(np.arccos(np.clip(c, -1, 1)))


# --------------------------------------------------------------------------------------------------------------------- tracking
def make_draws(gen, R, E, n_extra, S, n_pix_total):
    """one iteration's random inputs from a CPU generator: pixel indices + the sampler's draws (ray_sampler.py:58,148,158)"""
    return {"pix": torch.randint(n_pix_total, (R,), generator=gen),
            "t_rand": torch.rand(R, E, generator=gen), "extra_idx": torch.randperm(E, generator=gen)[:n_extra],
            "eik_idx": torch.randint(S, (R,), generator=gen)}


def track_sequence(engine, model, frames, K, gt_poses, H, W, iters=100, pixels=1024, lr=0.005, shared_seed=None, n_frames=None,
                   init_poses=None, log=None, trace=None, const_speed=False):
    """The reference's per-frame tracking protocol (volsdf_train.py:373-446) on `engine` in {"fused", "composed", "oracle"}.
    frames [n, H*W, 3] (on the model's device; CPU for the oracle).  shared_seed: every iteration's pixels and sampler draws come
    from a CPU generator seeded by (shared_seed, frame, iteration) -- identical for every engine; None: the engine's own RNG.
    init_poses: use these estimates for frames 0 .. len-1 (experiment B continues the oracle from the fused run's start).
    const_speed: SLAM.tracking.const_speed_assumption (volsdf_train.py:32,380-387; False in every shipped conf).
    trace: a list that receives (frame, iteration, loss, camera 7-vector BEFORE the step, camera gradient) of every iteration.
    -> est c2w [n,4,4] (CPU float32)."""
    from nicer_slam_amd.utils.general import get_camera_from_tensor, get_tensor_from_camera
    n = n_frames or frames.shape[0]
    oracle = engine == "oracle"
    dev = torch.device("cpu") if oracle else model.voxels.device
    samp = model.ray_sampler
    E, NX, S = samp.N_samples_eval, samp.N_samples_extra, samp.N_samples + 2 + samp.N_samples_extra
    if oracle:
        from oracle import render_ref as R
        mk = R.make_grid_spec
        ce = model.rendering_network.encoding
        cfg = R.RenderConfig(coarse=R.SdfNetSpec(mk(4, 8, 32, 32, 19), 2), fine=R.SdfNetSpec(mk(8, 4, 32, 128, 19), 4),
                             colour_grid=mk(ce.num_levels, ce.level_dim, ce.base_resolution, None, ce.log2_hashmap_size,
                                            per_level_scale=float(ce.per_level_scale)),
                             n_samples=samp.N_samples, n_samples_eval=E, n_samples_extra=NX)
        assert torch.equal(cfg.colour_grid.offsets.cpu(), ce.offsets.cpu().to(torch.int32)), "oracle grid layout != the model's"
        params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        vox = model.voxels.detach().cpu()
    else:
        model.engine = engine
        model.train()
    Kd = K.to(dev)
    est = [gt_poses[0].clone()]
    if init_poses is not None:
        est = [p.clone() for p in init_poses]
    ind = torch.zeros(1, dtype=torch.long, device=dev)
    gen_own = torch.Generator(device=dev).manual_seed(12345)
    for f in range(len(est), n):
        # initial estimate (volsdf_train.py:380-387): the previous frame's pose, or the constant-speed extrapolation
        if const_speed and f >= 2:
            c2w0 = (est[f - 1] @ torch.linalg.inv(est[f - 2])) @ est[f - 1]
        else:
            c2w0 = est[f - 1]
        cam = get_tensor_from_camera(c2w0.cpu()).to(dev).detach().clone().requires_grad_(True)
        opt = torch.optim.Adam([cam], lr=lr)
        sched = torch.optim.lr_scheduler.StepLR(opt, step_size=50, gamma=0.95)
        best, cand = float("inf"), None
        for it in range(iters):
            if shared_seed is not None:
                g = torch.Generator().manual_seed(shared_seed * 1000003 + f * 1009 + it)
                d = make_draws(g, pixels, E, NX, S, H * W)
                pix = d.pop("pix").to(dev)
                draws = {k: v.to(dev) for k, v in d.items()}
            else:
                pix = torch.randint(H * W, (pixels,), device=dev, generator=gen_own)
                draws = None
            uv = torch.stack([(pix % W).float(), (pix // W).float()], -1).unsqueeze(0)
            gt = frames[f].index_select(0, pix)
            if oracle:
                pose = R.camera_from_tensor(cam).unsqueeze(0)
                out = R.render(params, cfg, uv, pose, Kd[None], vox, draws, mode="tracking", training=True)
            else:
                model.draws = draws
                pose = get_camera_from_tensor(cam).unsqueeze(0)
                out = model({"intrinsics": Kd[None], "uv": uv, "pose": pose}, ind, {}, mode="tracking", frame_idx=f)
                assert model.last_engine == engine, model.last_engine
            loss = (out["rgb_values"].reshape(-1, 3) - gt).abs().mean()          # SLAMLoss tracking objective (loss.py:57-65,131)
            loss.backward()
            if trace is not None:
                trace.append((f, it, float(loss.detach()), cam.detach().cpu().clone(), cam.grad.detach().cpu().clone()))
            opt.step()
            sched.step()
            opt.zero_grad()
            lv = float(loss.detach())
            if lv < best:                                                      # :436-438 (the camera AFTER the step, as there)
                best, cand = lv, cam.detach().clone()
        est.append(get_camera_from_tensor(cand).detach().cpu().float() if not oracle else R.camera_from_tensor(cand).detach().float())
        if log is not None:
            log(f, est[-1], best)
    if not oracle:
        model.draws = None
    return torch.stack([e.cpu() for e in est])


# ------------------------------------------------------------------------------------------------- tracking AND mapping (mini SLAM)
def render_cues(model, poses, K, H, W):
    """colour, depth and (camera-frame) normal images [n, H*W, C] of the teacher: the synthetic stand-ins for a dataset's RGB frames
    and its monocular depth / normal cues (scene_dataset.py:185-205)."""
    from nicer_slam_amd.inference import render_image
    dev = model.voxels.device
    uv = all_pixels(H, W, dev).unsqueeze(0)
    was = model.training
    model.eval()
    rgb, depth, normal = [], [], []
    with torch.no_grad():
        for c2w in poses:
            res = render_image(model, {"intrinsics": K[None], "uv": uv, "pose": c2w.to(dev)[None]}, n_pixels=65536)
            rgb.append(res["rgb_values"].reshape(H * W, 3).clone())
            depth.append(res["depth_values"].reshape(H * W, 1).clone())
            normal.append(res["normal_map"].reshape(H * W, 3).clone())
    model.train(was)
    return torch.stack(rgb), torch.stack(depth), torch.stack(normal)


def render_analytic_room(poses, K, H, W, dev, half=(0.62, 0.5, 0.56)):
    """Frames of a scene NO network of this repository rendered: the inside of an axis-aligned box room with a smooth procedural
    texture on its walls, seen along `poses` -- colour, depth and camera-frame normal images [n, H*W, C] by closed-form ray / box
    intersection in the reference's ray conventions (directions divided by their squared norm, depth_values = depth_scale * z:
    rend_util.py:68-93, network.py:99-102,147-151; the normal is the inward one, the sign grad sdf has for inside_outside = true).
    The stand-in of rows g for an engine-independent data set: both engines (and any renderer) are handed the same images."""
    from nicer_slam_amd.utils import rend_util
    uv = all_pixels(H, W, dev).unsqueeze(0)
    Kd = K.to(dev)[None]
    hb = torch.tensor(half, device=dev)
    ds = rend_util.get_camera_params(uv, torch.eye(4, device=dev)[None], Kd)[0][0, :, 2:]           # depth_scale [HW,1]
    freq = torch.tensor([[3.1, 1.7, 2.3], [1.3, 4.1, 2.9], [2.2, 2.6, 5.3]], device=dev)             # cycles per unit, per colour channel
    phase = torch.tensor([0.3, 1.1, 2.0], device=dev)
    rgb, depth, normal = [], [], []
    for c2w in poses:
        c2w = c2w.to(dev)
        d, o = rend_util.get_camera_params(uv, c2w[None], Kd)
        d, o = d[0], o[0]
        assert bool((o.abs() < hb).all()), "the camera must be inside the room"
        t = torch.where(d > 0, (hb - o) / d, (-hb - o) / d)
        t = torch.where(d == 0, torch.full_like(t, 1e10), t)
        z, axis = t.min(dim=1)
        p = o + z[:, None] * d
        n_world = torch.zeros_like(d)
        n_world[torch.arange(d.shape[0], device=dev), axis] = -torch.sign(d[torch.arange(d.shape[0], device=dev), axis])
        tex = 0.5 + 0.25 * torch.sin(2 * np.pi * (p @ freq.t()) * 0.5 + phase) + 0.2 * torch.sin(2 * np.pi * p.sum(-1, keepdim=True) * 1.5 + phase)
        rgb.append(tex.clamp(0.02, 0.98))
        depth.append(z[:, None] * ds)
        normal.append(n_world @ c2w[:3, :3])                                                          # R^T n, as network.py:343-345 forms it
    return torch.stack(rgb), torch.stack(depth), torch.stack(normal)


def make_student(teacher, H, W, n_images, colour_grid=None, seed=11, family="replica"):
    """The map to be learned: a SLAMNetwork at the reference's initialisation (tables U(-1e-4, 1e-4), geometric MLPs) -- except the
    fine SDF MLP, which the reference loads from `pretrain.pth` and never optimises (volsdf_train.py:139-173): here the teacher's."""
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import model_conf
    torch.manual_seed(seed)
    kw = {} if colour_grid is None else {"colour_grid": colour_grid}
    student = SLAMNetwork(model_conf(family, 64, 640, 32, use_warp_loss=False), dataset=_DS(H, W), n_images=n_images, **kw)
    sd = teacher.state_dict()
    with torch.no_grad():
        for name, p in student.named_parameters():
            if name.startswith("implicit_network.fine.lin"):
                p.copy_(sd[name].to(p.device))
    student.train_dataset, student.keyframe_every = None, 10
    return student.to(teacher.voxels.device)


def run_slam(engine, teacher, rgb, depth, normal, K, gt, H, W, frames, colour_grid=None, map_every=5, map_iters=100, track_iters=100,
             map_pixels=8192, track_pixels=1024, lr=0.002, cam_lr=0.005, ba_lr=0.001, window=15, log=None, schedule="reference", seed=11,
             family="replica", none_grad="skip", const_speed=False):
    """Tracking AND mapping in the reference's loop shape (volsdf_train.py:363-613) on `engine`: frame 0 at its ground-truth pose and
    `map_iters` mapping iterations on it; every later frame tracked from the constant-speed initialisation against the map learned so far;
    every `map_every`-th frame a mapping round over the keyframe window (every 10th frame + the current one; the frames since the last
    keyframe join half-way), coarse -> fine and base -> highfreq schedules, bundle adjustment of the window's cameras in the last 30 % of a
    round (poses written back as :584-594 does).  Objective: the shipped SLAMLoss weights without the warp / flow terms; the monocular
    depth / normal cues are the teacher's renderings.  schedule = "reference": stage / colour-stage schedule of volsdf_train.py:550-555;
    "fine": every round at stage "fine" / "highfreq" -- on this synthetic scene the coarse-only quarter of a later round is BISTABLE
    (tools/diag_slam_mapping.py, profiles/r05_slam_mapping_rounds.txt: from one and the same state the round ends at loss 0.044 or loses the
    surface -- no ray straddles it any more, loss 0.14 -- on EITHER engine, by the draws), so trajectories under it compare luck, not engines.
    family: the conf family (model subtree + loss weights, utils/conf.py).  none_grad: the mapping optimizer's treatment of a table outside its
    stage -- "skip" = the installed torch (zero_grad() sets .grad = None, the table is not stepped), "zeros" = torch 1.11, the reference's
    environment (zero_grad() leaves zero tensors, the table keeps moving along its momentum; nicer_slam_amd/optim.py).
    -> final pose estimates [frames,4,4] (CPU), seconds spent in (tracking, mapping)."""
    from nicer_slam_amd.feed import FrameFeed
    from nicer_slam_amd.model.loss import SLAMLoss
    from nicer_slam_amd.optim import Adam as HipAdam
    from nicer_slam_amd.utils.general import get_camera_from_tensor, get_tensor_from_camera
    dev = teacher.voxels.device
    from nicer_slam_amd.utils.conf import run_conf
    lw = run_conf(family)["loss"]
    student = make_student(teacher, H, W, frames, colour_grid, seed, family)   # (also seeds every random stream of the run: torch's
    student.engine = engine                                              #  generators and, through them, the fused sampler's Philox state)
    student.train()
    imp, rn = student.implicit_network, student.rendering_network
    para_list = [     # volsdf_train.py:150-173
        {"name": "encoding", "params": list(imp.fine.grid_parameters()), "lr": lr * 20.0},
        {"name": "encoding", "params": list(imp.coarse.grid_parameters()), "lr": lr * 20.0},
        {"name": "net", "params": list(rn.grid_parameters()), "lr": lr * 5.0},
        {"name": "net", "params": list(rn.mlp_parameters()), "lr": lr},
        {"name": "density", "params": list(student.density.parameters()), "lr": 2e-3},
        {"name": "coarse_mlp_parameters", "params": list(imp.coarse.mlp_parameters()), "lr": lr},
    ]
    para_list = [g for g in para_list if len(g["params"])]
    optimizer = HipAdam(para_list, betas=(0.9, 0.99), eps=1e-15, none_grad=none_grad)   # (torch.optim.Adam's arithmetic, one pass per tensor)
    # the family's loss block (runconf_replica_1.conf:45-57 / runconf_7scenes_1.conf:46-58) without the warp / flow terms
    loss_fn = SLAMLoss(model=student, rgb_loss=lw["rgb_loss"], assign_scale_shift_init=lw["assign_scale_shift_init"],
                       eikonal_weight=lw["eikonal_weight"], smooth_weight=lw["smooth_weight"], depth_weight=lw["depth_weight"],
                       normal_l1_weight=lw["normal_l1_weight"], normal_cos_weight=lw["normal_cos_weight"])
    tracking_loss = SLAMLoss(model=student, rgb_loss="torch.nn.L1Loss", eikonal_weight=0, smooth_weight=0, depth_weight=0,
                             normal_l1_weight=0, normal_cos_weight=0)
    feed = FrameFeed((H, W), device=dev, capacity=frames)
    t_track = t_map = 0.0

    def mapping(frame_idx):
        local = [0] if frame_idx == 0 else list(range(0, frame_idx, 10)) + [frame_idx]
        for it in range(map_iters):
            if frame_idx != 0 and it == map_iters // 2:                                       # :493-495
                local = sorted(set(local + list(range(frame_idx // 10 * 10, frame_idx))))
            kf = list(local)
            feed.change_sampling_idx(max(1, map_pixels // len(kf)))
            indices, model_input, ground_truth = feed.batch(kf, full="store")
            ba = frame_idx != 0 and it > int(map_iters * 0.7)
            if ba:
                cams = torch.stack([get_tensor_from_camera((gt[0] if k == 0 else feed.frames[k]["pose"]).cpu()) for k in kf])
                cams = cams.to(dev).requires_grad_(True)
                opt_ba = torch.optim.Adam([cams], lr=ba_lr)
                model_input["pose"] = get_camera_from_tensor(cams)
            optimizer.zero_grad()
            if frame_idx > 1 and schedule == "reference":
                stage = "coarse" if it < int(map_iters * 0.25) else "fine"
                color_stage = "base" if it < int(map_iters * 0.7) else "highfreq"
            else:
                stage, color_stage = "fine", "highfreq"
            out = student(model_input, indices, ground_truth, keyframe_list=kf, frame_idx=frame_idx, mode="mapping", stage=stage,
                          color_stage=color_stage, iter=it)
            assert student.last_engine == engine, student.last_engine
            loss = loss_fn(out, ground_truth, kf, frame_idx=frame_idx, stage=stage)["loss"]
            loss.backward()
            optimizer.step()
            if ba:
                opt_ba.step()
                poses = get_camera_from_tensor(cams.detach())
                for ii, k in enumerate(kf):                                                   # :584-594
                    if k != 0 and not (k in kf[: window // 2]):
                        feed.set_pose(k, poses[ii])
        return float(loss.detach())

    for f in range(frames):
        if f == 0:
            pose0 = gt[0]
        elif f >= 2 and const_speed:
            p1, p2 = feed.frames[f - 1]["pose"].cpu(), feed.frames[f - 2]["pose"].cpu()
            pose0 = (p1 @ torch.linalg.inv(p2)) @ p1
        else:
            pose0 = feed.frames[f - 1]["pose"].cpu()
        # the reference's loader hands the monocular depth cue over in its own units and scales it by 20 on the first frame
        # (loss.py:179-185, assign_scale): the cue is the true depth / 20
        feed.add_frame(f, rgb=rgb[f], depth=depth[f] / 20.0, normal=normal[f], gt_depth=depth[f], intrinsics=K, pose=pose0)
        if f > 0:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cam = get_tensor_from_camera(pose0).to(dev).detach().clone().requires_grad_(True)
            opt = torch.optim.Adam([cam], lr=cam_lr)
            sched = torch.optim.lr_scheduler.StepLR(opt, step_size=50, gamma=0.95)
            best, cand = float("inf"), None
            for it in range(track_iters):
                feed.change_sampling_idx(track_pixels)
                indices, model_input, ground_truth = feed.batch([f])
                model_input["pose"] = get_camera_from_tensor(cam).unsqueeze(0)
                out = student(model_input, indices, ground_truth, mode="tracking", frame_idx=f)
                loss = tracking_loss(out, ground_truth, stage="fine", frame_idx=f)["loss"]
                loss.backward()
                opt.step()
                sched.step()
                opt.zero_grad()
                lv = float(loss.detach())
                if lv < best:
                    best, cand = lv, cam.detach().clone()
            feed.set_pose(f, get_camera_from_tensor(cand).detach())
            torch.cuda.synchronize()
            t_track += time.perf_counter() - t0
            if log is not None:
                e0 = float((pose0[:3, 3] - gt[f][:3, 3]).norm())
                e1 = float((feed.frames[f]["pose"].cpu()[:3, 3] - gt[f][:3, 3]).norm())
                log(f, best, f"tracked: initial error {e0:.5f} -> {e1:.5f}")
        if f % map_every == 0:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            last = mapping(f)
            torch.cuda.synchronize()
            t_map += time.perf_counter() - t0
            if log is not None:
                log(f, last, "mapped")
    est = torch.stack([feed.frames[f]["pose"].detach().cpu() for f in range(frames)])
    del optimizer, student, feed
    torch.cuda.empty_cache()
    return est, t_track, t_map


def run_slam_table(frames=50, H=340, W=600, colour_grid=None, map_iters=100, track_iters=100, engines=("fused", "composed"), verbose=False,
                   schedule="reference", seeds=(11,), family="replica", none_grads=("skip",), const_speed=False, analytic=False):
    """The mini-SLAM table: ATE of tracking + mapping on the synthetic sequence, fused engine beside the composed one; `none_grads`: the
    optimizer semantics to run ("skip" = installed torch, "zeros" = the reference's torch 1.11)."""
    dev = torch.device("cuda", 0)
    teacher = build_teacher(H, W, colour_grid=colour_grid, device=dev, family=family)
    teacher.engine = "fused"
    K = intrinsics(H, W, dev, family)
    gt = load_trajectory(frames, family=family)
    scale = FAMILIES[family]["scale"]
    if analytic:        # frames of the closed-form box room; the "teacher" only lends its fine SDF MLP (the pretrain.pth stand-in)
        rgb, depth, normal = render_analytic_room(gt, K, H, W, dev)
    else:
        rgb, depth, normal = render_cues(teacher, gt, K, H, W)
    out = {"what": "synthetic tracking + mapping (mini SLAM): teacher-rendered frames and depth / normal cues along gt_replica_room0[:N]; the "
                   "map is LEARNED (student at the reference's initialisation, fine SDF MLP = the teacher's as pretrain.pth is), loop shape of "
                   "volsdf_train.py:363-613 (mapping every 5th frame, 100 iterations x 8192 pixels, BA in the last 30 %), ATE as eval_cam.py:43-105",
           "conf_family": family, "trajectory": os.path.basename(FAMILIES[family]["traj"]), "scene_scale_units_per_m": scale,
           "const_speed_assumption": const_speed,
           "frames_rendered_by": "closed-form ray / box intersection of a textured box room (no network)" if analytic else "the teacher network (fused engine)",
           "gt_image_stats": {"mean": float(rgb.mean()), "std_over_pixels": float(rgb.std(dim=1).mean())},
           "frames": frames, "image": [H, W], "map_iters": map_iters, "track_iters": track_iters, "schedule": schedule,
           "no_tracking_baseline": summarise(gt, gt[:1].repeat(frames, 1, 1), scale)}
    log = (lambda f, l, what: print(f"  frame {f}: loss {l:.5f}  {what}", file=sys.stderr)) if verbose else None
    for none_grad in none_grads:
        tag = "" if none_grad == "skip" else "_none_grad_" + none_grad
        for spec in engines:                     # "engine" or "engine:seed,seed,..." (the composed engine is ~11x slower: fewer seeds)
            eng, _, sd = spec.partition(":")
            runs = []
            for seed in ([int(x) for x in sd.split("+")] if sd else list(seeds)):
                if verbose:
                    print(f"engine {eng} seed {seed} none_grad {none_grad}", file=sys.stderr)
                t0 = time.perf_counter()
                est, t_track, t_map = run_slam(eng, teacher, rgb, depth, normal, K, gt, H, W, frames, colour_grid, map_iters=map_iters,
                                               track_iters=track_iters, log=log, schedule=schedule, seed=seed, family=family,
                                               none_grad=none_grad, const_speed=const_speed)
                runs.append(dict(summarise(gt, est, scale), seed=seed, wall_s=round(time.perf_counter() - t0, 1), tracking_s=round(t_track, 1),
                                 mapping_s=round(t_map, 1)))
            ates = [r["ate_rmse_scene_units"] for r in runs]
            out["slam_" + eng + tag] = {"optimizer_none_grad": none_grad, "runs": runs, "ate_rmse_mean": float(np.mean(ates)),
                                        "ate_rmse_min": float(np.min(ates)), "ate_rmse_max": float(np.max(ates)),
                                        "ate_rmse_cm_at_room_scale_mean": float(np.mean(ates)) / scale * 100}
        if "slam_fused" + tag in out and "slam_composed" + tag in out:
            out["slam_ate_ratio_fused_over_composed" + tag] = out["slam_fused" + tag]["ate_rmse_mean"] / out["slam_composed" + tag]["ate_rmse_mean"]
    return out


def summarise(gt, est, scale=SCALE):
    gt, est = gt[:est.shape[0]].numpy(), est.numpy()
    tr = np.linalg.norm(gt[:, :3, 3] - est[:, :3, 3], axis=1)
    return {"frames": int(est.shape[0]), "ate_rmse_scene_units": ate_rmse(gt, est), "ate_rmse_cm_at_room_scale": ate_rmse(gt, est) / scale * 100,
            "mean_trans_err_scene_units": float(tr[1:].mean()), "max_trans_err_scene_units": float(tr[1:].max()),
            "mean_rot_err_deg": float(rot_err_deg(gt[1:, :3, :3], est[1:, :3, :3]).mean())}


def pose_diff(a, b):
    a, b = a.numpy(), b.numpy()
    n = min(len(a), len(b))
    return {"max_trans_diff_scene_units": float(np.linalg.norm(a[:n, :3, 3] - b[:n, :3, 3], axis=1).max()),
            "max_rot_diff_deg": float(rot_err_deg(a[:n, :3, :3], b[:n, :3, :3]).max())}


def trace_diff(a, b, first=5):
    """per-iteration agreement of two traces over frame 1 (same start, same draws): iteration 0 compares one forward + backward of
    the two engines; the camera drift over the first iterations shows how fast Adam amplifies fp32-level gradient differences"""
    a = [t for t in a if t[0] == 1][:first]
    b = [t for t in b if t[0] == 1][:first]
    n = min(len(a), len(b))
    if n == 0:
        return {}
    g0 = float((a[0][4] - b[0][4]).abs().max() / b[0][4].abs().max())
    return {"iter0_loss_rel_diff": abs(a[0][2] - b[0][2]) / abs(b[0][2]), "iter0_grad_diff_of_largest_component": g0,
            "max_cam_diff_first_%d_iters" % n: max(float((a[i][3] - b[i][3]).abs().max()) for i in range(n)),
            "max_loss_rel_diff_first_%d_iters" % n: max(abs(a[i][2] - b[i][2]) / abs(b[i][2]) for i in range(n))}


def run(frames=50, iters=100, pixels=1024, H=340, W=600, oracle_frames=3, oracle_pixels=128, oracle_iters=100, colour_grid=None,
        with_free=True, verbose=False, family="replica", const_speed=False, with_bf16=True, n_samples=64):
    dev = torch.device("cuda", 0)
    t_all = time.perf_counter()
    teacher = build_teacher(H, W, n_samples=n_samples, colour_grid=colour_grid, device=dev, family=family)
    teacher.engine = "fused"
    K = intrinsics(H, W, dev, family)
    gt = load_trajectory(frames, family=family)
    scale = FAMILIES[family]["scale"]
    imgs = render_frames(teacher, gt, K, H, W)
    speed = float(np.linalg.norm(np.diff(gt[:, :3, 3].numpy(), axis=0), axis=1).mean())
    out = {"what": "synthetic multi-frame tracking: teacher-rendered frames along gt_replica_room0[:N], reference tracking protocol "
                   "(volsdf_train.py:373-446), ATE as eval_cam.py:43-105",
           "conf_family": family, "trajectory": os.path.basename(FAMILIES[family]["traj"]), "const_speed_assumption": const_speed,
           "samples_per_ray": n_samples + 34,
           "frames": frames, "iters_per_frame": iters, "pixels_per_iter": pixels, "image": [H, W], "scene_scale_units_per_m": scale,
           "mean_frame_to_frame_motion_scene_units": speed,
           "gt_image_stats": {"mean": float(imgs.mean()), "std_over_pixels": float(imgs.std(dim=1).mean())}}
    log = (lambda f, p, b: print(f"  frame {f}: loss {b:.5f}", file=sys.stderr)) if verbose else None
    # reference point: what a tracker that does nothing would score (every frame = frame 0)
    out["no_tracking_baseline"] = summarise(gt, gt[:1].repeat(frames, 1, 1), scale)
    if with_free:
        res = {}
        for eng in ("fused", "composed"):
            t0 = time.perf_counter()
            est = track_sequence(eng, teacher, imgs, K, gt, H, W, iters, pixels, log=log, const_speed=const_speed)
            torch.cuda.synchronize()
            res[eng] = est
            out["free_running_" + eng] = dict(summarise(gt, est, scale), wall_s=round(time.perf_counter() - t0, 1),
                                              ms_per_iteration=round((time.perf_counter() - t0) / ((frames - 1) * iters) * 1e3, 3))
        a, b = out["free_running_fused"]["ate_rmse_scene_units"], out["free_running_composed"]["ate_rmse_scene_units"]
        out["free_running_ate_ratio_fused_over_composed"] = a / b
        out["free_running_pose_difference"] = pose_diff(res["fused"], res["composed"])
        # the optional reduced-precision modes of BASELINE configs[2] / [4] on the same frames (fp32 teacher, bf16-operand tracker)
        for prec in (("bf16", "bf16_colour") if with_bf16 else ()):
            teacher.mlp_precision = prec
            teacher.__dict__.pop("_track_graphs", None)
            t0 = time.perf_counter()
            est = track_sequence("fused", teacher, imgs, K, gt, H, W, iters, pixels, log=log, const_speed=const_speed)
            torch.cuda.synchronize()
            out["free_running_fused_" + prec] = dict(summarise(gt, est, scale), wall_s=round(time.perf_counter() - t0, 1),
                                                     ms_per_iteration=round((time.perf_counter() - t0) / ((frames - 1) * iters) * 1e3, 3))
        teacher.mlp_precision = "fp32"
        teacher.__dict__.pop("_track_graphs", None)
    # (B) shared draws, reduced pixel count so that the CPU oracle can take part
    nB = max(2, oracle_frames + 1)
    estB, trB = {}, {}
    for eng in ("fused", "composed"):
        trB[eng] = []
        estB[eng] = track_sequence(eng, teacher, imgs, K, gt, H, W, oracle_iters, oracle_pixels, shared_seed=7, n_frames=nB, trace=trB[eng],
                                   const_speed=const_speed)
        out["shared_draws_" + eng] = summarise(gt, estB[eng], scale)
    out["shared_draws_fused_vs_composed"] = dict(pose_diff(estB["fused"], estB["composed"]), **trace_diff(trB["fused"], trB["composed"]))
    if oracle_frames > 0:
        t0 = time.perf_counter()
        teacher_cpu = teacher.to("cpu")
        trB["oracle"] = []
        estB["oracle"] = track_sequence("oracle", teacher_cpu, imgs.cpu(), K.cpu(), gt, H, W, oracle_iters, oracle_pixels, shared_seed=7,
                                        n_frames=nB, trace=trB["oracle"], const_speed=const_speed)
        out["shared_draws_oracle"] = dict(summarise(gt, estB["oracle"], scale), wall_s=round(time.perf_counter() - t0, 1))
        out["shared_draws_fused_vs_oracle"] = dict(pose_diff(estB["fused"], estB["oracle"]), **trace_diff(trB["fused"], trB["oracle"]))
        out["shared_draws_composed_vs_oracle"] = dict(pose_diff(estB["composed"], estB["oracle"]), **trace_diff(trB["composed"], trB["oracle"]))
    out["shared_draws"] = {"frames": nB, "iters_per_frame": oracle_iters, "pixels_per_iter": oracle_pixels,
                           "note": "identical pixels and sampler draws: iteration 0 of frame 1 compares one forward + backward; later "
                                   "iterations compound fp32-level gradient differences through Adam (lr 0.005 = 1.7 frame steps per "
                                   "unit m/sqrt(v)) and the arg-min-loss candidate picks an ITERATION, which is discontinuous in the "
                                   "losses -- final poses of two engines differ at the level of the tracker's own scatter"}
    out["wall_s"] = round(time.perf_counter() - t_all, 1)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--pixels", type=int, default=1024)
    ap.add_argument("--conf", default="replica", choices=sorted(FAMILIES), help="conf family: model subtree, loss weights, camera, trajectory")
    ap.add_argument("--height", type=int, default=None, help="default: the family's frame size halved (340 x 600 / 240 x 320)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--const-speed", default="conf", choices=["conf", "on", "off"],
                    help="SLAM.tracking.const_speed_assumption; 'conf' = the family's shipped value (false in all 23 files)")
    ap.add_argument("--none-grad", default="skip", help="comma list of optimizer semantics for --slam: skip (installed torch), zeros (torch 1.11)")
    ap.add_argument("--seeds", default="11")
    ap.add_argument("--n-samples", type=int, default=64, help="N_samples of the ray sampler (composite samples per ray = this + 34); tracking table only")
    ap.add_argument("--analytic", action="store_true", help="--slam on frames of a closed-form textured box room instead of teacher renderings")
    ap.add_argument("--oracle-frames", type=int, default=3)
    ap.add_argument("--oracle-pixels", type=int, default=128)
    ap.add_argument("--oracle-iters", type=int, default=100)
    ap.add_argument("--small-colour-grid", action="store_true", help="64 MiB colour table instead of the shipped 1 GiB (quick runs)")
    ap.add_argument("--no-free", action="store_true")
    ap.add_argument("--slam", action="store_true", help="the tracking + mapping table (map learned from the frames) instead of the tracking one")
    ap.add_argument("--map-iters", type=int, default=100)
    ap.add_argument("--engines", default="fused,composed")
    ap.add_argument("--schedule", default="reference", choices=["reference", "fine"])
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    cg = dict(base_resolution=16, desired_resolution=512, log2_hashmap_size=19) if a.small_colour_grid else None
    from nicer_slam_amd.utils.conf import run_conf
    H, W = a.height or FAMILIES[a.conf]["size"][0], a.width or FAMILIES[a.conf]["size"][1]
    cs = run_conf(a.conf)["const_speed_assumption"] if a.const_speed == "conf" else a.const_speed == "on"
    if a.slam:
        print(json.dumps(run_slam_table(a.frames, H, W, cg, a.map_iters, a.iters, tuple(a.engines.split(",")), a.verbose, a.schedule,
                                        seeds=tuple(int(x) for x in a.seeds.split(",")), family=a.conf,
                                        none_grads=tuple(a.none_grad.split(",")), const_speed=cs, analytic=a.analytic), indent=1))
        sys.exit(0)
    print(json.dumps(run(a.frames, a.iters, a.pixels, H, W, a.oracle_frames, a.oracle_pixels, a.oracle_iters, cg,
                         not a.no_free, a.verbose, family=a.conf, const_speed=cs, n_samples=a.n_samples), indent=1))
