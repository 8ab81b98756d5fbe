"""The two tilings of the SDF-network kernels -- quad (16 points per wave, four lanes per point: csrc/render_sdfnet4.hip,
render_sampler4.hip; the default) and 32-point (lane pair per point: csrc/render_sdfnet.hip, render_sampler.hip) -- compute
the same function: identical inputs, every output and every gradient compared.  (All other GPU tests run on the default
tiling and hold it to the oracle / the reference goldens; this file keeps the 32-point kernels held to the same numbers.)"""
import numpy as np
import pytest
import torch

from helpers import load, tt, draws_of, golden_objective, assert_close
from test_model_cpu import build_model

pytestmark = pytest.mark.gpu


def _run(fx, tile, z=None):
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    model = build_model(fx).cuda()
    model.engine = "fused"
    model.sdf_tile = tile
    model.tracking_param_grads = True
    model.fine_mlp_grads = True
    mode, stage, cstage = str(fx["meta_mode"]), str(fx["meta_stage"]), str(fx["meta_color_stage"])
    model.train(True)
    model.voxels = tt(fx["in_voxels"]).cuda()
    model.draws = draws_of(fx, "cuda")
    if z is not None:
        model.draws["z_vals_override"] = z
    cam = tt(fx["in_cam"]).cuda().requires_grad_(True)
    out = model({"intrinsics": tt(fx["in_K"]).cuda(), "uv": tt(fx["in_uv"]).cuda(), "pose": get_camera_from_tensor(cam)},
                torch.arange(cam.shape[0], device="cuda"), {}, mode=mode, stage=stage, color_stage=cstage, frame_idx=1)
    assert model.last_engine == "fused"
    golden_objective(out, fx, mode).backward()
    return out, cam.grad, {n: p.grad for n, p in model.named_parameters()}


@pytest.mark.parametrize("name", ["full_tracking_rw", "full_mapping_rw", "full_mapping_rw_coarse", "full_tracking_poisson",
                                  "full_tracking_7scenes", "full_mapping_7scenes"])
def test_quad_and_32_point_tilings_agree(name):
    fx = load(name)
    o16, c16, g16 = _run(fx, 16)
    z = o16["z_vals"].detach()
    o32, c32, g32 = _run(fx, 32, z)
    o16, c16, g16 = _run(fx, 16, z)
    # free-running sampler of the quad tiling vs the reference's samples is covered by tests/test_sampler_gpu.py
    for k in ("sdf", "weights", "rgb", "rgb_values", "depth_values", "normal_map", "entropy", "grad_theta", "grad_theta_nei"):
        if k in o16:
            # (weights and what is composited with them: d sigma / d sdf = 1 / (2 beta^2) ~ 2400 amplifies the last-ulp
            # differences of the two summation orders)
            per_point = k in ("sdf", "rgb", "grad_theta", "grad_theta_nei")
            assert_close(o16[k], o32[k].detach().cpu().numpy(), 5e-6 if per_point else 2e-5, 1e-5, k)
    assert_close(c16, c32.cpu().numpy(), 1e-7 + 1e-4 * float(c32.abs().max()), 1e-4, "grad_cam")
    n = 0
    for k, g in g32.items():
        if g is None:
            assert g16[k] is None, k
            continue
        assert_close(g16[k], g.cpu().numpy(), 1e-8 + 2e-4 * float(g.abs().max()), 1e-3, "grad " + k)
        n += 1
    assert n >= 10


def _ws_tiles():
    """Tile codes 96 / 97 (the wave-specialised experiment samplers) exist only in a tagged side-by-side build made
    with NSA_X_WS=1 and loaded through NSA_LIB_TAG; the product library refuses them (test_product_library_refuses_ws_codes)."""
    from nicer_slam_amd._native import lib
    return (96, 97) if hasattr(lib, "nsa_sampler_ws_sdf") else ()


def test_product_library_refuses_ws_codes():
    from nicer_slam_amd._native import lib
    if hasattr(lib, "nsa_sampler_ws_sdf"):
        pytest.skip("experiment build with the wave-specialised samplers loaded")
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.fused import sampler as fs
    model = SLAMNetwork(replica_model_conf(94, 640, 32, use_warp_loss=False), n_images=1,
                        colour_grid=dict(base_resolution=16, desired_resolution=64, log2_hashmap_size=12)).cuda().train()
    o = torch.zeros(4, 3, device="cuda")
    d = torch.nn.functional.normalize(torch.ones(4, 3, device="cuda"), dim=-1)
    for tile in (96, 97):
        model.sdf_tile = tile
        with pytest.raises(RuntimeError):
            fs.sampler_sdf(model, o, d, torch.rand(4, 640, device="cuda"))


def test_sampler_sdf_stage_both_tilings():
    """Coarse sampler stage (z, sdf at R*E points) and batch SDF inference, quad vs 32-point tiling, shipped grid sizes, ragged
    point counts (the quad kernels are persistent: every tile must be visited exactly once)."""
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.fused import sampler as fs
    from nicer_slam_amd import inference
    torch.manual_seed(0)
    model = SLAMNetwork(replica_model_conf(94, 640, 32, use_warp_loss=False), n_images=1,
                        colour_grid=dict(base_resolution=16, desired_resolution=64, log2_hashmap_size=12)).cuda().train()
    g = torch.Generator(device="cuda").manual_seed(3)
    with torch.no_grad():
        for enc in (model.implicit_network.coarse.encoding, model.implicit_network.fine.encoding):
            enc.embeddings.copy_((torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1) * 0.05)
        for n_, p in model.named_parameters():
            if n_.startswith("implicit_network") and n_.endswith("weight_v"):
                p.add_(0.05 * torch.randn(p.shape, device="cuda", generator=g))
    for R in (1000, 37):
        d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda", generator=g), dim=-1) * 0.7
        o = (torch.rand(R, 3, device="cuda", generator=g) - 0.5) * 0.4
        t_rand = torch.rand(R, 640, device="cuda", generator=g)
        res = {}
        # 64 = 32-point tiling, two point tiles per wave; 96 = wave-specialised 32-point form (render_sampler_ws.hip)
        # 97 = systolic form (layer-engine waves, render_sampler_sys.hip)
        for tile in (16, 32, 64) + _ws_tiles():
            model.sdf_tile = tile
            res[tile] = fs.sampler_sdf(model, o, d, t_rand)
        for a, b, what in zip(res[16], res[32], ("z", "sdf", "far")):
            assert_close(a, b.cpu().numpy(), 1e-6 if what == "sdf" else 0, 1e-5 if what == "sdf" else 0, f"{what} (R={R})")
        for a, b, what in zip(res[64], res[32], ("z", "sdf", "far")):
            assert torch.equal(a, b), f"two tiles per wave: {what} differs from one tile per wave (R={R})"
        for t in _ws_tiles():                                            # same MFMA order per accumulator: bit-identical
            for a, b, what in zip(res[t], res[32], ("z", "sdf", "far")):
                assert torch.equal(a, b), f"wave-specialised sampler (tile code {t}): {what} differs from the one-program form " \
                                          f"(R={R}): {int((a != b).sum())} of {a.numel()}"
    pts = (torch.rand(100003, 3, device="cuda", generator=g) * 2 - 1) * 1.2
    for stage in ("fine", "coarse"):
        vals = {}
        for tile in (16, 32):
            model.sdf_tile = tile
            vals[tile] = inference.sdf_values(model, pts, stage, chunk=30011)
        assert_close(vals[16], vals[32].cpu().numpy(), 1e-6, 1e-5, "sdf_values " + stage)


def test_mfma_kernels_are_bit_reproducible_and_match_the_oracle_at_scale():
    """Regression for the packed-fp32 (SLP v_pk_*_f32) miscompile/hazard found in round 2 (nicer_slam_amd/build.py): with it,
    the same binary on the same inputs returned different sdf values at ~1e-4 of 640 k points (errors to 3e-3), invisible while
    the first layer ignored its positional-encoding / grid columns (geometric initialisation).  Random first-layer weights, shipped
    grid sizes: five runs of each kernel must agree bitwise, and a random subset must equal the CPU oracle."""
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.fused import sampler as fs
    from nicer_slam_amd import inference
    from oracle import render_ref as R
    torch.manual_seed(0)
    model = SLAMNetwork(replica_model_conf(94, 640, 32, use_warp_loss=False), n_images=1,
                        colour_grid=dict(base_resolution=16, desired_resolution=64, log2_hashmap_size=12)).cuda().train()
    g = torch.Generator(device="cuda").manual_seed(3)
    with torch.no_grad():
        for enc in (model.implicit_network.coarse.encoding, model.implicit_network.fine.encoding):
            enc.embeddings.copy_((torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1) * 0.05)
        for n_, p in model.named_parameters():
            if n_.startswith("implicit_network") and n_.endswith("weight_v"):
                p.add_(0.05 * torch.randn(p.shape, device="cuda", generator=g))
    Rn = 1000
    d = torch.nn.functional.normalize(torch.randn(Rn, 3, device="cuda", generator=g), dim=-1) * 0.7
    o = (torch.rand(Rn, 3, device="cuda", generator=g) - 0.5) * 0.4
    t_rand = torch.rand(Rn, 640, device="cuda", generator=g)
    mk = R.make_grid_spec
    cfg = R.RenderConfig(coarse=R.SdfNetSpec(mk(4, 8, 32, 32, 19), 2), fine=R.SdfNetSpec(mk(8, 4, 32, 128, 19), 4),
                         colour_grid=mk(16, 2, 16, 64, 12), n_samples=94, n_samples_eval=640, n_samples_extra=32)
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    for tile in (16, 32) + _ws_tiles():
        model.sdf_tile = tile
        runs = [fs.sampler_sdf(model, o, d, t_rand) for _ in range(5)]
        for r in runs[1:]:
            assert torch.equal(r[1], runs[0][1]), f"sampler (tile {tile}) is not reproducible: {int((r[1] != runs[0][1]).sum())} points"
        z, sdf = runs[0][0], runs[0][1]
        pts = (o.unsqueeze(1) + z.unsqueeze(2) * d.unsqueeze(1)).reshape(-1, 3).contiguous()
        vals = [inference.sdf_values(model, pts, "fine") for _ in range(5)]
        for v in vals[1:]:
            assert torch.equal(v, vals[0]), f"sdf_points (tile {tile}) is not reproducible"
        assert_close(vals[0], sdf.reshape(-1).cpu().numpy(), 1e-6, 1e-5, "sampler vs sdf_points")
        sel = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(1))[:40000]
        with torch.no_grad():
            ref = R.sdf_vals(params, cfg, pts[sel.cuda()].cpu()).reshape(-1)
        assert_close(sdf.reshape(-1)[sel.cuda()], ref, 2e-5, 1e-4, f"sampler sdf vs oracle (tile {tile})")


@pytest.mark.parametrize("mode,P", [("rays", 64 * 128), ("rays", 37 * 98), ("points", 1000), ("points", 5)])
def test_paired_forward_is_the_two_quad_launches_bit_for_bit(mode, P):
    """nsa_sdfnet_forward_pair (both networks of the COMBINE in one launch, coarse results carried in registers) against
    nsa_sdfnet_forward(coarse, accumulate 0) + nsa_sdfnet_forward(fine, accumulate 1) in the quad tiling: sdf, grad sdf and the
    HL feature buffer must be identical -- shipped grid sizes, ragged point counts, ray-sample and explicit-point sources, and a
    permuted launch order."""
    import ctypes
    from nicer_slam_amd._native import lib, check, PointsDesc
    from nicer_slam_amd.fused import render as fr, sampler as fs
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    torch.manual_seed(0)
    model = SLAMNetwork(replica_model_conf(use_warp_loss=False), n_images=1,
                        colour_grid=dict(base_resolution=16, desired_resolution=64, log2_hashmap_size=12)).cuda().train()
    g = torch.Generator(device="cuda").manual_seed(7)
    with torch.no_grad():
        for enc in (model.implicit_network.coarse.encoding, model.implicit_network.fine.encoding):
            enc.embeddings.copy_((torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1) * 0.05)
        for n_, p in model.named_parameters():
            if n_.startswith("implicit_network") and n_.endswith("weight_v"):
                p.add_(0.05 * torch.randn(p.shape, device="cuda", generator=g))
    model.sdf_tile = 16
    assert fs.forward_pair_ok(model)
    gc, keep_c = fs.sdf_grid_desc(model, "coarse", "coarse_pair")
    gf, keep_f = fs.sdf_grid_desc(model, "fine")
    pc, pf = fs.packed_sdf(model, "coarse", use="coarse_pair"), fs.packed_sdf(model, "fine")
    if mode == "rays":
        S = 128 if P % 128 == 0 else 98
        R = P // S
        rays_d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda", generator=g), dim=-1)
        rays_o = ((torch.rand(R, 3, device="cuda", generator=g) - 0.5) * 0.4).contiguous()
        z = torch.sort(torch.rand(R, S, device="cuda", generator=g) * 1.6, dim=1).values.contiguous()     # some samples leave the cube
        srcs = [(PointsDesc(rays_o.data_ptr(), rays_d.data_ptr(), z.data_ptr(), None, P, S, None), None)]
        order = torch.randperm(P, device="cuda", generator=g).to(torch.int32)
        srcs.append((PointsDesc(rays_o.data_ptr(), rays_d.data_ptr(), z.data_ptr(), None, P, S, order.data_ptr()), order))
    else:
        x = ((torch.rand(P, 3, device="cuda", generator=g) * 2 - 1) * 1.05).contiguous()
        srcs = [(PointsDesc(None, None, None, x.data_ptr(), P, 0, None), None)]
    st = torch.cuda.current_stream().cuda_stream
    for pts, _ in srcs:
        two = [torch.full((P,), float("nan"), device="cuda"), torch.full((P, 3), float("nan"), device="cuda"),
               torch.zeros(fr.hl_size(P), device="cuda")]
        one = [t.clone() for t in two]
        check(lib.nsa_sdfnet_forward(ctypes.byref(pts), ctypes.byref(gc), pc.data_ptr(), 0, two[0].data_ptr(), two[1].data_ptr(),
                                     two[2].data_ptr(), st))
        check(lib.nsa_sdfnet_forward(ctypes.byref(pts), ctypes.byref(gf), pf.data_ptr(), 1, two[0].data_ptr(), two[1].data_ptr(),
                                     two[2].data_ptr(), st))
        check(lib.nsa_sdfnet_forward_pair(ctypes.byref(pts), ctypes.byref(gc), ctypes.byref(gf), pc.data_ptr(), pf.data_ptr(),
                                          one[0].data_ptr(), one[1].data_ptr(), one[2].data_ptr(), st))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(two[0]).all()) and float(two[0].abs().max()) > 0
        for a, b, what in zip(one, two, ("sdf", "grad sdf", "features")):
            assert torch.equal(a, b), f"{what}: {int((a != b).sum())} of {a.numel()} differ, max {float((a - b).abs().max()):.3g}"
    # mismatched descriptors are refused
    g32, _k = fs.grid_desc(model.implicit_network.coarse.encoding, model.implicit_network.coarse.divide_factor, 1, 0, 32)
    assert lib.nsa_sdfnet_forward_pair(ctypes.byref(srcs[0][0]), ctypes.byref(g32), ctypes.byref(gf), pc.data_ptr(), pf.data_ptr(),
                                       one[0].data_ptr(), one[1].data_ptr(), one[2].data_ptr(), st) == 4


@pytest.mark.parametrize("R,S,stage,color_stage", [(64, 128, "fine", "highfreq"), (37, 98, "fine", "base"), (5, 64, "coarse", "highfreq")])
def test_colour_plus_coarse_backward_in_one_launch_changes_no_bit(R, S, stage, color_stage):
    """nsa_colour_coarse_backward (the colour backward and the coarse SDF backward of the same 32-point tiles as two phases of one launch)
    against nsa_colour_backward + nsa_sdfnet_backward(coarse, accumulate 1): the same statements, so every output of the data-path
    backward -- d/dx, d/d(view dir), the feature and normal cotangents and the ray sums -- must be identical; ragged point counts,
    both stages, colour grid gradient on and off."""
    from nicer_slam_amd.fused import render as fr
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    torch.manual_seed(0)
    model = SLAMNetwork(replica_model_conf(use_warp_loss=False), n_images=1,
                        colour_grid=dict(base_resolution=16, desired_resolution=128, log2_hashmap_size=13)).cuda().train()
    for p in model.parameters():
        p.requires_grad_(False)
    g = torch.Generator(device="cuda").manual_seed(5)
    with torch.no_grad():
        for enc in (model.implicit_network.coarse.encoding, model.implicit_network.fine.encoding, model.rendering_network.encoding):
            enc.embeddings.copy_((torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1) * 0.05)
        for n_, p in model.named_parameters():
            if n_.endswith("weight_v"):
                p.add_(0.05 * torch.randn(p.shape, device="cuda", generator=g))
    rays_d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda", generator=g), dim=-1)
    rays_o = ((torch.rand(R, 3, device="cuda", generator=g) - 0.5) * 0.4).contiguous()
    z = torch.sort(torch.rand(R, S, device="cuda", generator=g) * 1.4 + 0.05, dim=1).values.contiguous()
    g_rgbv = torch.randn(R, 3, device="cuda", generator=g)
    g_depth = torch.randn(R, device="cuda", generator=g)
    outs = {}
    assert fr.COLOUR_COARSE_BWD
    for merged in (False, True):
        fr.COLOUR_COARSE_BWD = merged
        try:
            b = fr.composite_forward_raw(model, rays_o, rays_d, z, stage, True)
            b["_keep"] = {}
            g_o, g_d = fr.composite_backward_raw(model, rays_o, rays_d, z, b, stage, color_stage, g_rgbv=g_rgbv, g_depth=g_depth)
        finally:
            fr.COLOUR_COARSE_BWD = True
        torch.cuda.synchronize()
        k = b["_keep"]
        outs[merged] = dict(g_o=g_o, g_d=g_d, g_x=k["g_x"], g_dir=k["g_dir"], g_feat=k["g_feat"], g_grad=k["g_grad"])
    assert bool(torch.isfinite(outs[False]["g_x"]).all()) and float(outs[False]["g_x"].abs().max()) > 0
    live = fr.hl_index(R * S, "cuda")                  # [P,64] positions of the live points' features in the HL buffer: the columns of the
    for name in outs[False]:                           # last tile's dead lanes derive from never-written forward features (garbage in, ignored)
        a, b_ = outs[True][name], outs[False][name]
        if name == "g_feat":
            a, b_ = a[live], b_[live]
        assert torch.equal(a, b_), f"{name}: {int((a != b_).sum())} of {a.numel()} differ, max {float((a - b_).abs().max()):.3g}"


def test_colour_forward_with_the_composite_in_its_launch_changes_no_bit():
    """nsa_colour_forward_composite (128 samples per ray: the colour forward's workgroup is one ray and its first wave runs the
    composite forward when the colours are stored) against nsa_colour_forward + nsa_composite_forward: weights, rendered colour, depth,
    normal map, entropy and the per-sample colours must be identical."""
    from nicer_slam_amd.fused import render as fr
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    torch.manual_seed(0)
    model = SLAMNetwork(replica_model_conf(use_warp_loss=False), n_images=1,
                        colour_grid=dict(base_resolution=16, desired_resolution=128, log2_hashmap_size=13)).cuda().train()
    for p in model.parameters():
        p.requires_grad_(False)
    g = torch.Generator(device="cuda").manual_seed(9)
    with torch.no_grad():
        for enc in (model.implicit_network.coarse.encoding, model.implicit_network.fine.encoding, model.rendering_network.encoding):
            enc.embeddings.copy_((torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1) * 0.05)
        for n_, p in model.named_parameters():
            if n_.endswith("weight_v"):
                p.add_(0.05 * torch.randn(p.shape, device="cuda", generator=g))
    R, S = 77, 128
    rays_d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda", generator=g), dim=-1)
    rays_o = ((torch.rand(R, 3, device="cuda", generator=g) - 0.5) * 0.4).contiguous()
    z = torch.sort(torch.rand(R, S, device="cuda", generator=g) * 1.4 + 0.05, dim=1).values.contiguous()
    outs = {}
    assert fr.COLOUR_FWD_TRACK
    for merged in (False, True):
        fr.COLOUR_FWD_TRACK = merged
        try:
            b = fr.composite_forward_raw(model, rays_o, rays_d, z, "fine", True)
        finally:
            fr.COLOUR_FWD_TRACK = True
        torch.cuda.synchronize()
        outs[merged] = {k: b[k].clone() for k in ("weights", "rgb_values", "depth", "nmap", "entropy", "rgb", "sdf")}
    assert bool(torch.isfinite(outs[False]["rgb_values"]).all()) and float(outs[False]["weights"].abs().max()) > 0
    for k in outs[False]:
        a, b_ = outs[True][k], outs[False][k]
        assert torch.equal(a, b_), f"{k}: {int((a != b_).sum())} of {a.numel()} differ, max {float((a - b_).abs().max()):.3g}"

