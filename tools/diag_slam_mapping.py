#!/usr/bin/env python
"""Diagnostic for the mini-SLAM of tools/synthetic_sequence.py: ONE mapping round at frame 5 (keyframes [0, 5], the frames in between
joining half-way; coarse -> fine, base -> highfreq, BA in the last 30 %) started from the SAME model / optimizer / pose state on the fused
and on the composed engine, per-iteration loss terms side by side.  Found in round 5: which engine's mapping round differs, and where."""
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synthetic_sequence as ss


def main(H=170, W=300, frames=6, map_iters=100, map_pixels=8192, variants=("fused", "composed"), first_engine="fused"):
    from nicer_slam_amd.feed import FrameFeed
    from nicer_slam_amd.model.loss import SLAMLoss
    from nicer_slam_amd.optim import Adam as HipAdam
    from nicer_slam_amd.utils.general import get_camera_from_tensor, get_tensor_from_camera
    dev = torch.device("cuda", 0)
    teacher = ss.build_teacher(H, W, device=dev)
    teacher.engine = "fused"
    K = ss.intrinsics(H, W, dev)
    gt = ss.load_trajectory(frames)
    rgb, depth, normal = ss.render_cues(teacher, gt, K, H, W)
    student = ss.make_student(teacher, H, W, frames)
    student.train()
    lr = 0.002
    imp, rn = student.implicit_network, student.rendering_network

    def make_opt(model):
        i, r = model.implicit_network, model.rendering_network
        groups = [{"params": list(i.fine.grid_parameters()), "lr": lr * 20.0}, {"params": list(i.coarse.grid_parameters()), "lr": lr * 20.0},
                  {"params": list(r.grid_parameters()), "lr": lr * 5.0}, {"params": list(r.mlp_parameters()), "lr": lr},
                  {"params": list(i.coarse.mlp_parameters()), "lr": lr}]
        return HipAdam(groups, betas=(0.9, 0.99), eps=1e-15)

    def make_loss(model):
        return SLAMLoss(model=model, rgb_loss="torch.nn.L1Loss", assign_scale_shift_init=True, eikonal_weight=0.1, smooth_weight=0.005,
                        depth_weight=0.1, normal_l1_weight=0.05, normal_cos_weight=0.05)

    feed = FrameFeed((H, W), device=dev, capacity=frames)
    for f in range(frames):                      # ground-truth poses + a small error: the state a tracker would leave
        pose = gt[f].clone()
        if f > 0:
            pose[:3, 3] += 0.002 * torch.tensor([1.0, -1.0, 0.5]) * (1 if f % 2 else -1)
        feed.add_frame(f, rgb=rgb[f], depth=depth[f] / 20.0, normal=normal[f], gt_depth=depth[f], intrinsics=K, pose=pose)

    def mapping(model, optimizer, loss_fn, engine, frame_idx, iters, trace):
        model.engine = engine
        local = [0] if frame_idx == 0 else list(range(0, frame_idx, 10)) + [frame_idx]
        poses0 = {k: feed.frames[k]["pose"].clone() for k in feed.frames}
        for it in range(iters):
            if frame_idx != 0 and it == iters // 2:
                local = sorted(set(local + list(range(frame_idx // 10 * 10, frame_idx))))
            kf = list(local)
            feed.change_sampling_idx(max(1, map_pixels // len(kf)))
            indices, model_input, ground_truth = feed.batch(kf, full="store")
            ba = frame_idx != 0 and it > int(iters * 0.7)
            if ba:
                cams = torch.stack([get_tensor_from_camera((gt[0] if k == 0 else feed.frames[k]["pose"]).cpu()) for k in kf]).to(dev).requires_grad_(True)
                opt_ba = torch.optim.Adam([cams], lr=0.001)
                model_input["pose"] = get_camera_from_tensor(cams)
            optimizer.zero_grad()
            if frame_idx > 1:
                stage = "coarse" if it < int(iters * 0.25) else "fine"
                color_stage = "base" if it < int(iters * 0.7) else "highfreq"
            else:
                stage, color_stage = "fine", "highfreq"
            out = model(model_input, indices, ground_truth, keyframe_list=kf, frame_idx=frame_idx, mode="mapping", stage=stage,
                        color_stage=color_stage, iter=it)
            terms = loss_fn(out, ground_truth, kf, frame_idx=frame_idx, stage=stage)
            terms["loss"].backward()
            optimizer.step()
            if ba:
                opt_ba.step()
                poses = get_camera_from_tensor(cams.detach())
                for ii, k in enumerate(kf):
                    if k != 0 and not (k in kf[:7]):
                        feed.set_pose(k, poses[ii])
            if it % 5 == 0 or it == iters - 1:
                trace.append((it, stage, color_stage, len(kf), {k: float(v) for k, v in terms.items() if torch.is_tensor(v) or isinstance(v, float)}))
        for k, p in poses0.items():                # put the poses back for the next variant
            feed.set_pose(k, p)

    # frame 0 round on the fused engine: the common starting point
    opt0, loss0 = make_opt(student), make_loss(student)
    tr0 = []
    mapping(student, opt0, loss0, first_engine, 0, map_iters, tr0)
    print(f"frame-0 round ({first_engine}):", " ".join(f"{t[0]}:{t[4]['loss']:.4f}" for t in tr0[::4]))
    with torch.no_grad():
        v = student.voxels
        print(f"  state after it: visit counter sum {float(v.sum()):.6g} max {float(v.max()):.6g} nonzero {int((v > 0).sum())};  "
              + "  ".join(f"{n.split('.')[-3] if 'lin' in n else n.split('.')[1]}.{n.split('.')[-1]} |{float(p.norm()):.5g}|"
                          for n, p in student.named_parameters() if n.endswith(("embeddings", "lin0.weight_v", "lin2.weight_g"))))
    state = copy.deepcopy(student.state_dict())
    opt_state = copy.deepcopy(opt0.state_dict())
    vox = student.voxels.clone()
    for eng in variants:
        student.load_state_dict(state)
        with torch.no_grad():
            student.voxels.copy_(vox)
        student.__dict__.pop("_fused_pack", None)
        opt = make_opt(student)
        opt.load_state_dict(copy.deepcopy(opt_state))
        tr = []
        t0 = time.perf_counter()
        mapping(student, opt, make_loss(student), eng, 5, map_iters, tr)
        torch.cuda.synchronize()
        print(f"\n== frame-5 round on {eng} ({time.perf_counter() - t0:.1f} s)")
        for it, stage, cs, nk, terms in tr:
            print(f"  it {it:3d} {stage:6s} {cs:8s} kf {nk}  " + "  ".join(f"{k} {v:.5f}" for k, v in terms.items()))


if __name__ == "__main__":
    main(first_engine=sys.argv[1] if len(sys.argv) > 1 else "fused")
