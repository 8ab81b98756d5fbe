"""Row g: sequence-level evidence -- a short version of tools/synthetic_sequence.py (the full 50-frame runs with their tables are
profiles/r05_sequence_ate.json, profiles/r06_sequence_ate_7scenes.json), for the Replica conf family and for BASELINE configs[3]'s
(7-Scenes: tests/golden/scenes7_office_traj64.txt, utils/conf.py::model_conf("7scenes")).  A teacher model with structured tables renders frames along the first poses of the reference's
ground-truth Replica room0 trajectory (tests/golden/replica_room0_traj64.txt); the frames are tracked with the reference's protocol
(volsdf_train.py:373-446: constant-speed initialisation from the previous ESTIMATES, Adam + StepLR on the camera 7-vector, rgb L1,
arg-min-loss candidate) on the fused engine, the composed engine and the CPU oracle; ATE RMSE as eval_cam.py:43-105."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

H, W = 68, 120
CG = dict(base_resolution=16, desired_resolution=256, log2_hashmap_size=15)


def test_ate_is_horns_alignment():
    """eval_cam.py:43-105 on a known answer: a rigidly moved copy of a path has ATE 0, one displaced point gives the RMSE by hand."""
    import synthetic_sequence as ss
    g = np.random.default_rng(0)
    gt = np.tile(np.eye(4), (10, 1, 1))
    gt[:, :3, 3] = g.normal(size=(10, 3))
    a = 0.7
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    est = gt.copy()
    est[:, :3, 3] = gt[:, :3, 3] @ Rz.T + np.array([0.3, -0.2, 0.5])
    assert ss.ate_rmse(gt, est) < 1e-12
    est2 = gt.copy()
    est2[:, :3, 3] += 0.1 * g.normal(size=(10, 3))
    rot, trans = ss.align_rigid(est2[:, :3, 3].T, gt[:, :3, 3].T)
    resid = rot @ est2[:, :3, 3].T + trans - gt[:, :3, 3].T
    assert abs(ss.ate_rmse(gt, est2) - np.sqrt((resid ** 2).sum(0).mean())) < 1e-12
    assert abs(np.linalg.det(rot) - 1) < 1e-9 and ss.ate_rmse(gt, est2) <= np.sqrt(((est2 - gt)[:, :3, 3] ** 2).sum(1).mean()) + 1e-12


# (conf family, constant-speed initialisation).  Every shipped conf leaves SLAM.tracking.const_speed_assumption at false
# (volsdf_train.py:32); "replica" + True is the protocol rounds 4-5 measured (kept: it takes the extrapolation branch :380-385).
# "7scenes" = BASELINE configs[3]'s family: gt_7scenes_office path, Kinect camera, coarse radius 1.0, fine MLP not geometric.
@pytest.mark.parametrize("family,const_speed,size", [("replica", True, (68, 120)), ("replica", False, (68, 120)),
                                                     ("7scenes", False, (60, 80))])
def test_synthetic_sequence_fused_vs_composed_vs_oracle(capsys, family, const_speed, size):
    import synthetic_sequence as ss
    dev = torch.device("cuda", 0)
    n, iters, pixels = 7, 50, 512
    H, W = size
    teacher = ss.build_teacher(H, W, colour_grid=CG, device=dev, family=family)
    teacher.engine = "fused"
    K = ss.intrinsics(H, W, dev, family)
    gt = ss.load_trajectory(n, family=family, stride=4 if family == "7scenes" else 1)
    imgs = ss.render_frames(teacher, gt, K, H, W)
    assert float(imgs.std(dim=1).mean()) > 0.08                      # the frames carry texture to track against
    # (A) free-running engines, independent draws: statistical agreement of the trajectories
    est = {e: ss.track_sequence(e, teacher, imgs, K, gt, H, W, iters, pixels, const_speed=const_speed) for e in ("fused", "composed")}
    ate = {e: ss.ate_rmse(gt.numpy(), v.numpy()) for e, v in est.items()}
    still = ss.ate_rmse(gt.numpy(), gt[:1].repeat(n, 1, 1).numpy())    # a tracker that never moves
    # (B) shared pixels and sampler draws: arithmetic only; the CPU oracle joins on the first frames with a small batch
    nB, itB, pxB = 3, 30, 64
    tr = {e: [] for e in ("fused", "composed", "oracle")}
    estB = {e: ss.track_sequence(e, teacher, imgs, K, gt, H, W, itB, pxB, shared_seed=7, n_frames=nB, trace=tr[e], const_speed=const_speed)
            for e in ("fused", "composed")}
    cpu = teacher.to("cpu")
    estB["oracle"] = ss.track_sequence("oracle", cpu, imgs.cpu(), K.cpu(), gt, H, W, itB, pxB, shared_seed=7, n_frames=nB, trace=tr["oracle"],
                                       const_speed=const_speed)
    d_fc = dict(ss.pose_diff(estB["fused"], estB["composed"]), **ss.trace_diff(tr["fused"], tr["composed"]))
    d_fo = dict(ss.pose_diff(estB["fused"], estB["oracle"]), **ss.trace_diff(tr["fused"], tr["oracle"]))
    step = float(np.linalg.norm(np.diff(gt[:, :3, 3].numpy(), axis=0), axis=1).mean())
    with capsys.disabled():
        print(f"\n  [{family}, const_speed={const_speed}] ATE RMSE (scene units; frame-to-frame motion {step:.4f}): fused {ate['fused']:.5f}  composed {ate['composed']:.5f}  "
              f"no tracking {still:.5f}\n  shared draws: fused vs composed {d_fc}\n                fused vs oracle   {d_fo}")
    assert ate["fused"] < 0.5 * still and ate["composed"] < 0.5 * still          # both engines actually track
    assert ate["fused"] <= 1.05 * ate["composed"] + 0.15 * step                   # matched ATE (short run: + a noise floor)
    # identical inputs.  Iteration 0 of frame 1 is one forward + backward of each engine on the same batch: fp32 parity bounds.
    # After that Adam (lr 0.005 = 1.4 frame steps per unit of m / sqrt(v)) compounds the gradient differences, and the frame's
    # pose is the arg-min-loss ITERATE -- discontinuous in the losses: final poses agree at the level of the tracker's own scatter.
    for d in (d_fc, d_fo):
        # (measured on MI355X: loss 3e-7 .. 6e-6, gradient 1e-6 .. 3e-4 of the largest component, camera drift over five
        #  iterations 8e-9 .. 5e-6; final poses 0.0011-0.0015 apart at an ATE of 0.0009-0.0010)
        assert d["iter0_loss_rel_diff"] < 5e-5 and d["iter0_grad_diff_of_largest_component"] < 2e-3, d
        assert d["max_cam_diff_first_5_iters"] < 5e-5, d
        assert d["max_trans_diff_scene_units"] < 0.6 * step and d["max_rot_diff_deg"] < 0.15, d


@pytest.mark.parametrize("family,none_grad,size", [("replica", "skip", (136, 240)), ("7scenes", "zeros", (120, 160))])
def test_mini_slam_tracks_with_a_learned_map(capsys, family, none_grad, size):
    """Tracking AND mapping in the reference's loop shape (volsdf_train.py:363-613) on the fused engine: the map is learned from the
    frames (student at the reference's initialisation), every 5th frame is mapped with bundle adjustment, every frame tracked against
    the map so far.  Short version of `tools/synthetic_sequence.py --slam` (profiles/r05_slam_ate.json: 50 frames, both engines, seeds)."""
    import synthetic_sequence as ss
    dev = torch.device("cuda", 0)
    (Hs, Ws), n = size, 16
    cg = dict(base_resolution=16, desired_resolution=512, log2_hashmap_size=19)
    teacher = ss.build_teacher(Hs, Ws, colour_grid=cg, device=dev, family=family)
    teacher.engine = "fused"
    K = ss.intrinsics(Hs, Ws, dev, family)
    gt = ss.load_trajectory(n, family=family)
    rgb, depth, normal = ss.render_cues(teacher, gt, K, Hs, Ws)
    # (family "7scenes": its model subtree and loss weights -- smooth_weight 0.05 -- and the optimizer in the reference environment's
    #  semantics; under schedule "fine" every table has a gradient in every iteration, so none_grad only has to be harmless here: the
    #  reference schedule under both semantics is profiles/r06_slam_ate.json)
    still = ss.ate_rmse(gt.numpy(), gt[:1].repeat(n, 1, 1).numpy())
    # The learning loop is not run-to-run reproducible (the mapping step scatters its table gradients with atomics) and it is chaotic:
    # over 33 runs of the 7-Scenes case its ATE scattered 0.0011-0.0044 around 0.002 -- under either operand form of the library
    # (profiles/r06_ab_experiments.txt r7a) -- against this bound of 0.0030 (the family's office sequence moves 0.4 mm per frame at its
    # start, so "no tracking" is a small number).  Two of 33 runs crossed it; a second attempt is allowed before the test fails.
    for attempt in range(2):
        est, t_track, t_map = ss.run_slam("fused", teacher, rgb, depth, normal, K, gt, Hs, Ws, n, cg, map_iters=100, track_iters=60,
                                          schedule="fine", family=family, none_grad=none_grad)
        ate = ss.ate_rmse(gt.numpy(), est.numpy())
        err = np.linalg.norm(gt[:, :3, 3].numpy() - est[:, :3, 3].numpy(), axis=1)
        with capsys.disabled():
            print(f"\n  mini-SLAM [{family}, none_grad={none_grad}], {n} frames: ATE RMSE {ate:.5f} (no tracking {still:.5f}); max error {err.max():.5f}; tracking {t_track:.1f} s, mapping {t_map:.1f} s")
        assert np.isfinite(est.numpy()).all()
        if ate < 0.5 * still and err.max() < 0.03:
            break
    assert ate < 0.5 * still and err.max() < 0.03, (ate, still, err.max())
