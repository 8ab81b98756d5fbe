"""BASELINE.json configs at their exact workload shapes on the HIP path, checked against the CPU oracle.

configs[0]  "demo_2 ... 256 rays x 64 samples -- plumbing/ref": R = 256, S = 64 (N_samples = 30 + near + far + 32 extras),
            E = 640 sampler evaluations per ray, shipped grid sizes (coarse 32^3 x4 levels x8, fine 32->128 x8 levels x4,
            colour 16->2048 x16 levels x2 = 1 GiB), fp32, one tracking iteration: forward dict + pose gradient.
The oracle needs ~1 s for this batch, so it IS the checker here (configs[1] at 1024 x 128 is covered by the size-independent
properties of tests/test_properties_gpu.py and by the goldens)."""
import pytest
import torch

from helpers import assert_close
from test_oracle_golden import check_samples

pytestmark = pytest.mark.gpu


class _DS:
    img_res = (680, 1200)


@pytest.mark.parametrize("poisson", [False, True])
def test_config0_256_rays_64_samples_vs_oracle(poisson):
    _check_shape(256, 64, poisson)


@pytest.mark.parametrize("Rn,S,what", [(128, 128, "configs[1] / [2]: 128 samples per ray (the bench shape), reduced ray count"),
                                       (96, 192, "configs[4]: 192 samples per ray, reduced ray count")])
def test_other_config_sample_counts_vs_oracle(Rn, S, what):
    """The per-ray kernels' shapes depend on S (LDS tiles, wave-scan lengths, sort width), not on the ray count: the remaining
    BASELINE sample counts at a ray count the oracle renders in about a second."""
    _check_shape(Rn, S, False)


def _check_shape(Rn, S, poisson):
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    from oracle import render_ref as R
    E, NX = 640, 32
    torch.manual_seed(0)
    model = SLAMNetwork(replica_model_conf(S - 2 - NX, E, NX, use_warp_loss=False), dataset=_DS(), n_images=1).cuda().train()
    g = torch.Generator(device="cuda").manual_seed(3)
    with torch.no_grad():
        for enc, s in ((model.implicit_network.coarse.encoding, 0.02), (model.implicit_network.fine.encoding, 0.02),
                       (model.rendering_network.encoding, 0.3)):
            enc.embeddings.copy_((torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1) * s)
    for p in model.parameters():
        p.requires_grad_(False)
    if poisson:       # second run with Poisson(50) visit counts (beta varies per voxel)
        model.voxels = torch.poisson(torch.full((64, 64, 64), 50.0, device="cuda"), generator=g)
    model.engine = "fused"
    idx = torch.randint(680 * 1200, (1, Rn), device="cuda", generator=g)
    uv = torch.stack([(idx % 1200).float(), (idx // 1200).float()], -1)
    K = torch.eye(4, device="cuda")
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    gt = torch.rand(Rn, 3, device="cuda", generator=g)
    draws = {"t_rand": torch.rand(Rn, E, device="cuda", generator=g),
             "extra_idx": torch.randperm(E, device="cuda", generator=g)[:NX],
             "eik_idx": torch.randint(S, (Rn,), device="cuda", generator=g)}
    model.draws = dict(draws)
    cam = torch.tensor([1.0, 0.01, -0.02, 0.015, 0.1, 0.0, -0.2], device="cuda", requires_grad=True)
    inp = {"intrinsics": K[None], "uv": uv, "pose": get_camera_from_tensor(cam).unsqueeze(0)}
    out = model(inp, torch.zeros(1, dtype=torch.long, device="cuda"), {}, mode="tracking", frame_idx=1)
    assert model.last_engine == "fused" and out["z_vals"].shape == (Rn, S)
    loss = (out["rgb_values"].reshape(-1, 3) - gt).abs().mean()
    loss.backward()

    mk = R.make_grid_spec
    cfg = R.RenderConfig(coarse=R.SdfNetSpec(mk(4, 8, 32, 32, 19), 2), fine=R.SdfNetSpec(mk(8, 4, 32, 128, 19), 4),
                         colour_grid=mk(16, 2, 16, 2048, 24), n_samples=S - 2 - NX, n_samples_eval=E, n_samples_extra=NX)
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    dc = {k: v.cpu() for k, v in draws.items()}
    vox = model.voxels.cpu()
    # (1) free-running oracle: its own sampler -> sample sets compared in CDF space (tests/test_oracle_golden.py)
    cam_c = cam.detach().cpu().clone().requires_grad_(True)
    free = R.render(params, cfg, uv.cpu(), R.camera_from_tensor(cam_c).unsqueeze(0), K[None].cpu(), vox, dict(dc),
                    mode="tracking", training=True)
    check_samples(out["z_vals"].cpu(), free["z_vals"], free["sampler_bins"], free["sampler_cdf"], u_tol=5e-5)
    for k in ("rgb_values", "depth_values", "normal_map"):
        assert_close(out[k], free[k], 2e-4, 1e-3, "free-running " + k)
    # (2) everything downstream of the sampler, tight, from a FIXED sample set on both sides: the GPU's samples with the far
    # sample pulled 2e-4 inside.  As drawn, the far sample sits exactly ON the cube face, where every grid's in-range test
    # (hashencoder.cu:155-159) hangs on the last ulp of o + z d; the two sides build their rays with differently ordered fp32
    # sums (as the reference on CUDA vs on CPU would), so that one sample may be inside on one side and outside on the other
    #  -- with a finest-level Jacobian of ~600/unit that shows in the pose gradient, not in the rendered values.
    z_fix = out["z_vals"].detach().clone()
    z_fix[:, -1] = torch.maximum(z_fix[:, -1] * (1 - 2e-4), z_fix[:, -2])
    model.draws = dict(draws, z_vals_override=z_fix)
    cam2 = cam.detach().clone().requires_grad_(True)
    out = model({"intrinsics": K[None], "uv": uv, "pose": get_camera_from_tensor(cam2).unsqueeze(0)},
                torch.zeros(1, dtype=torch.long, device="cuda"), {}, mode="tracking", frame_idx=1)
    assert model.last_engine == "fused"
    dc["z_vals_override"] = z_fix.cpu()
    cam_c = cam.detach().cpu().clone().requires_grad_(True)
    ref = R.render(params, cfg, uv.cpu(), R.camera_from_tensor(cam_c).unsqueeze(0), K[None].cpu(), vox, dc,
                   mode="tracking", training=True)
    for k in ("sdf", "depth_vals", "rgb", "weights", "rgb_values", "depth_values", "normal_map", "entropy"):
        assert_close(out[k], ref[k], 2e-5, 1e-4, k)
    # Pose gradient.  The colour MLP's ReLUs make the reference's own gradient discontinuous: a unit whose pre-activation is
    # ~1e-7 (typical: 0.1) at some point takes a different mask in two fp32 evaluations that agree to the last digits (here:
    # one point in 16k, unit 44 of layer 0 at 7e-7, found with tools/diag_config0.py -- 18 % of that point's gradient).  Rays
    # holding a point within 2e-6 of a kink are left out of the objective ON BOTH SIDES; their number is bounded.
    zc = z_fix.cpu()
    with torch.no_grad():
        pose_c = R.camera_from_tensor(cam.detach().cpu()).unsqueeze(0)
        d_c, o_c = R.camera_rays(uv.cpu(), pose_c, K[None].cpu())
        pts = (o_c.unsqueeze(1) + zc.unsqueeze(2) * d_c.reshape(-1, 3).unsqueeze(1)).reshape(-1, 3)
    _, feat_c, _ = R.sdf_outputs(params, cfg, pts.clone(), "fine")
    margin = R.colour_relu_margin(params, cfg, pts, ref["gradients"].detach(), d_c.reshape(-1, 3).unsqueeze(1).repeat(1, S, 1).reshape(-1, 3),
                                  feat_c.detach())
    # ~2.1 M ReLU units per batch with a pre-activation density of ~2 per unit length around 0: ~17 sit within 2e-6 of the kink
    kink = (margin.reshape(Rn, S) < 2e-6).any(dim=1)
    assert int(kink.sum()) <= Rn // 6, int(kink.sum())
    keep = (~kink).float().unsqueeze(-1)
    l_ref = ((ref["rgb_values"].reshape(-1, 3) - gt.cpu()).abs() * keep).sum() / (3 * Rn)
    l_ref.backward()
    cam2.grad = None
    out = model({"intrinsics": K[None], "uv": uv, "pose": get_camera_from_tensor(cam2).unsqueeze(0)},
                torch.zeros(1, dtype=torch.long, device="cuda"), {}, mode="tracking", frame_idx=1)
    loss = ((out["rgb_values"].reshape(-1, 3) - gt).abs() * keep.cuda()).sum() / (3 * Rn)
    loss.backward()
    assert_close(loss, l_ref, 1e-6, 1e-5, "loss")
    assert_close(cam2.grad, cam_c.grad, 1e-3 * float(cam_c.grad.abs().max()), 1e-3, "pose gradient")
    # ... and component by component ("1e-3 of the largest component" says little about the small ones): every one of the
    # seven within 1e-3 of ITSELF plus a floor of 5e-5 of the largest (the ray sums of ~1e5 fp32 terms in two summation orders)
    g_h, g_c = cam2.grad.detach().cpu().double(), cam_c.grad.double()
    per = ((g_h - g_c).abs() / g_c.abs().clamp_min(1e-30)).tolist()
    print("pose gradient, relative error per component:", ["%.1e" % v for v in per], " components / max:",
          ["%.2f" % v for v in (g_c.abs() / g_c.abs().max()).tolist()])
    assert bool(((g_h - g_c).abs() <= 1e-3 * g_c.abs() + 5e-5 * g_c.abs().max()).all()), per


@pytest.mark.parametrize("Rn,S", [(256, 64), (1024, 128)])
def test_effects_the_tight_comparison_leaves_out_are_measured(Rn, S):
    """_check_shape() compares tightly after (a) pulling the far sample -- which sits exactly ON the cube face -- 2e-4 inside and
    (b) leaving rays with a colour-MLP ReLU unit within 2e-6 of its kink out of the objective.  Here NOTHING is pulled or masked,
    at configs[0] (256 x 64) and configs[1] (1024 x 128) size, and the effects are counted and bounded instead of removed:
      * far samples whose in-range decision (hashencoder.cu:155-159) differs between the rays of the HIP kernel and the
        oracle's torch rays, per grid;
      * rays holding a near-kink ReLU unit;
      * the resulting difference of the un-masked pose gradient (looser bound than the tight test's 1e-3, stated below);
      * the forward tensors at SURVEY 8c's tolerance, 1e-5 abs / 1e-4 rel, from the same (un-pulled) samples."""
    from nicer_slam_amd.fused import render as fused_render
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    from oracle import render_ref as R
    E, NX = 640, 32
    torch.manual_seed(0)
    model = SLAMNetwork(replica_model_conf(S - 2 - NX, E, NX, use_warp_loss=False), dataset=_DS(), n_images=1).cuda().train()
    g = torch.Generator(device="cuda").manual_seed(3)
    with torch.no_grad():
        for enc, s in ((model.implicit_network.coarse.encoding, 0.02), (model.implicit_network.fine.encoding, 0.02),
                       (model.rendering_network.encoding, 0.3)):
            enc.embeddings.copy_((torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1) * s)
    for p in model.parameters():
        p.requires_grad_(False)
    model.engine = "fused"
    idx = torch.randint(680 * 1200, (1, Rn), device="cuda", generator=g)
    uv = torch.stack([(idx % 1200).float(), (idx // 1200).float()], -1)
    K = torch.eye(4, device="cuda")
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    gt = torch.rand(Rn, 3, device="cuda", generator=g)
    draws = {"t_rand": torch.rand(Rn, E, device="cuda", generator=g),
             "extra_idx": torch.randperm(E, device="cuda", generator=g)[:NX],
             "eik_idx": torch.randint(S, (Rn,), device="cuda", generator=g)}
    model.draws = dict(draws)
    cam = torch.tensor([1.0, 0.01, -0.02, 0.015, 0.1, 0.0, -0.2], device="cuda", requires_grad=True)
    pose = get_camera_from_tensor(cam).unsqueeze(0)
    out = model({"intrinsics": K[None], "uv": uv, "pose": pose}, torch.zeros(1, dtype=torch.long, device="cuda"), {},
                mode="tracking", frame_idx=1)
    ((out["rgb_values"].reshape(-1, 3) - gt).abs().mean()).backward()
    z = out["z_vals"].detach()

    # ---- (a) far samples on the cube face: in-range decision from the kernel's rays vs from the oracle's rays ----
    with torch.no_grad():
        o_h, d_h, _ = fused_render.rays(pose.detach(), uv, K[None])
        pose_c = R.camera_from_tensor(cam.detach().cpu()).unsqueeze(0)
        d_c, o_c = R.camera_rays(uv.cpu(), pose_c, K[None].cpu())
        far = z[:, -1:]
        x_h = (o_h.reshape(-1, 3) + far * d_h.reshape(-1, 3)).cpu()
        x_c = o_c.reshape(1, 3) + far.cpu() * d_c.reshape(-1, 3)
    on_face = ((x_c.abs().amax(-1) - 1.0).abs() < 4e-6)
    inside = lambda x: (((x + 1) / 2 >= 0) & ((x + 1) / 2 <= 1)).all(-1)        # divide_factor = 1 for all three grids
    flips = int((inside(x_h) != inside(x_c)).sum())
    assert float(on_face.float().mean()) > 0.9, "the far sample is the cube exit by construction (ray_sampler.py:23-35)"
    assert flips <= Rn // 4, flips                       # measured: a few per cent of the rays

    # ---- oracle on the SAME samples, nothing pulled, nothing masked ----
    mk = R.make_grid_spec
    cfg = R.RenderConfig(coarse=R.SdfNetSpec(mk(4, 8, 32, 32, 19), 2), fine=R.SdfNetSpec(mk(8, 4, 32, 128, 19), 4),
                         colour_grid=mk(16, 2, 16, 2048, 24), n_samples=S - 2 - NX, n_samples_eval=E, n_samples_extra=NX)
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    dc = {k: v.cpu() for k, v in draws.items()}
    dc["z_vals_override"] = z.cpu()
    cam_c = cam.detach().cpu().clone().requires_grad_(True)
    ref = R.render(params, cfg, uv.cpu(), R.camera_from_tensor(cam_c).unsqueeze(0), K[None].cpu(), model.voxels.cpu(), dc,
                   mode="tracking", training=True)
    R.rgb_l1(ref, gt.cpu()).backward()

    # ---- (b) rays with a ReLU unit of the colour MLP on its kink ----
    with torch.no_grad():
        pts = (o_c.unsqueeze(1) + z.cpu().unsqueeze(2) * d_c.reshape(-1, 3).unsqueeze(1)).reshape(-1, 3)
    _, feat_c, _ = R.sdf_outputs(params, cfg, pts.clone(), "fine")
    margin = R.colour_relu_margin(params, cfg, pts, ref["gradients"].detach(),
                                  d_c.reshape(-1, 3).unsqueeze(1).repeat(1, S, 1).reshape(-1, 3), feat_c.detach())
    kinks = int((margin.reshape(Rn, S) < 2e-6).any(dim=1).sum())
    assert kinks <= Rn // 6, kinks

    # ---- forward tensors at SURVEY 8c's 1e-5 abs / 1e-4 rel (per-sample quantities only where the far sample's grid
    # membership agrees: a flipped sample legitimately differs by its whole colour-grid feature) ----
    same = (inside(x_h) == inside(x_c))
    for k in ("rgb_values", "depth_values", "normal_map"):
        a, b = out[k].detach().cpu().reshape(Rn, -1)[same], ref[k].detach().reshape(Rn, -1)[same]
        assert_close(a, b, 1e-5, 1e-4, k)
    assert_close(out["sdf"].detach().cpu()[:, :-1], ref["sdf"].detach().reshape(Rn, S)[:, :-1], 1e-5, 1e-4, "sdf")
    assert_close(out["weights"].detach().cpu()[same], ref["weights"].detach()[same], 1e-5, 1e-4, "weights")

    # ---- the un-masked, un-pulled pose gradient: bounded, looser than the tight comparison's 1e-3 ----
    g_h, g_c = cam.grad.cpu(), cam_c.grad
    rel = float((g_h - g_c).abs().max() / g_c.abs().max())
    print(f"[measured] R={Rn} S={S}: far-sample in-range flips {flips}, kink rays {kinks}, "
          f"un-masked pose-gradient difference {rel:.2e} of its largest component")
    # measured on MI355X: 3.9e-4 (256 x 64: 8 flips, 19 kink rays) and 3.1e-4 (1024 x 128: 19 flips, 133 kink rays); a single ReLU
    # unit that does take different masks on the two sides moved it by 6.6e-3 in round 2 (tools/diag_config0.py)
    assert rel < 1e-2, rel
