"""HIP kernels of the keyframe re-projection blocks (C ABI section 5: nsa_patch_warp_*, nsa_flow_*, nsa_masked_l1;
csrc/warp_terms.hip, fused/warp.py) against (i) the function-level golden captured from the reference's forward
(reproj_blocks.npz: forward tensors, d/d depth and direct d/d pose of every term, total camera gradient through the fused
engine), (ii) the golden-pinned torch restatement (model/warp.py) at the shipped image size, patch sizes 1 / 5 / 11, with and
without bundle adjustment, with a resident frame store."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import assert_close, load, tt, draws_of
from test_warp_cpu import check_backward, check_forward, reproj_inputs

pytestmark = pytest.mark.gpu


def _kernel_blocks(model, d, index=None, store=None):
    from nicer_slam_amd.fused import warp as fw
    bs = d["uv"].shape[0]
    if store is not None:
        gt = {"full_rgb": fw.FrameStore(store[0], index), "full_depth": fw.FrameStore(store[1], index)}
    else:
        gt = {"full_rgb": d["full_rgb"], "full_depth": d["full_depth"]}
    warp_out = fw.patch_warp(model, d["uv"], d["pose"], d["K"], d["depth"], gt, bs)
    flow = fw.flow(model, d["uv"], d["pose"], d["K"], d["depth"], d["edges"])
    return warp_out, flow


def test_kernels_match_reference_reprojection_golden():
    from nicer_slam_amd.fused.warp import masked_l1
    fx = load("reproj_blocks")
    d, model = reproj_inputs(fx, "cuda")
    warp_out, flow = _kernel_blocks(model, d)
    check_forward(fx, warp_out, flow)
    terms = {f"warp{ps}": masked_l1(s, g, m, 3) for ps, (g, s, m, _) in warp_out.items()}
    terms["flow"] = masked_l1(flow, d["gt_flow"], d["flow_mask"], 2)
    for (tag, t), ref in zip(terms.items(), list(fx["out_warp_terms"]) + [fx["out_flow_term"]]):
        assert abs(float(t) - float(ref)) < 1e-5 * max(1.0, abs(float(ref))), tag
    check_backward(fx, terms, d["depth"], d["pose"])


def _random_batch(bs, n, seed, H=680, W=1200, frames=None):
    g = torch.Generator(device="cuda").manual_seed(seed)
    frames = frames or bs
    K = torch.eye(4, device="cuda")
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    idx = torch.randint(H * W, (bs, n), device="cuda", generator=g)
    uv = torch.stack([(idx % W).float(), (idx // W).float()], -1)
    uv[0, :4] = torch.tensor([[0.0, 0.0], [2.0, 3.0], [W - 1.0, H - 1.0], [W - 3.0, 4.0]], device="cuda")   # patches leave the image
    cam = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2], device="cuda").repeat(bs, 1)
    cam = cam + 0.02 * torch.randn(bs, 7, device="cuda", generator=g)
    # smooth frames + 2 % noise: |sampled - gt| and the bilinear sample are only piecewise differentiable (a projected coordinate
    # within the evaluation-order noise of an integer picks the neighbouring texel cell: same value, different slope); on
    # white-noise frames those kinks carry O(1) slope jumps and dominate the summed camera gradient of the 11 x 11 patches
    i = torch.arange(H * W, device="cuda")
    uu, vv = (i % W).float(), (i // W).float()
    ph = torch.rand(frames, 1, 3, device="cuda", generator=g) * 6.28
    rgb = 0.5 + 0.35 * torch.sin(uu[None, :, None] * 0.021 + ph) * torch.cos(vv[None, :, None] * 0.017 + 0.5 * ph)
    rgb = rgb + 0.02 * torch.rand(frames, H * W, 3, device="cuda", generator=g)
    dep = 1.0 + 0.05 * torch.rand(frames, H * W, 1, device="cuda", generator=g)
    dep[:, : H * W // 2] += 0.4 * torch.rand(frames, H * W // 2, 1, device="cuda", generator=g)
    depth = 0.5 + 2.0 * torch.rand(bs, n, device="cuda", generator=g)
    depth[0, 5] = -0.3                                                    # behind the camera
    return uv, K[None].repeat(bs, 1, 1), cam, rgb, dep, depth


@pytest.mark.parametrize("ba", [False, True])
@pytest.mark.parametrize("patches", [[1], [1, 5, 11]])
def test_kernels_vs_torch_twin_at_image_size(patches, ba):
    from nicer_slam_amd.fused import warp as fw
    from nicer_slam_amd.model.warp import flow_reproject, patch_warp
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    bs, n = 4, 300
    uv, K, cam, rgb, dep, depth0 = _random_batch(bs, n, 3, frames=6)
    index = torch.tensor([4, 0, 5, 2], device="cuda", dtype=torch.int32)
    model = SimpleNamespace(H=680, W=1200, patchsizes=patches)
    edges = (torch.tensor([0, 1, 2, 3, 0], device="cuda"), torch.tensor([1, 0, 3, 2, 3], device="cuda"), None, None)
    gt_flow = torch.randn(5, n, 2, device="cuda") * 20
    flow_mask = torch.rand(5, n, device="cuda") > 0.3
    res = {}
    for engine in ("torch", "hip"):
        cam_l = cam.clone().requires_grad_(ba)
        pose = get_camera_from_tensor(cam_l)
        depth = depth0.clone().requires_grad_(True)
        if engine == "torch":
            gt = {"full_rgb": rgb.index_select(0, index.long()), "full_depth": dep.index_select(0, index.long())}
            wo = patch_warp(model, uv, pose, K, depth.reshape(-1, 1).unsqueeze(2), gt, bs)
            fl = flow_reproject(uv, pose, K, depth, edges)
            terms = [(s[m] - g[m]).abs().mean() for (g, s, m, _) in wo.values()]
            terms.append((fl[flow_mask] - gt_flow[flow_mask]).abs().mean())
        else:
            d = dict(uv=uv, pose=pose, K=K, depth=depth, edges=edges)
            wo, fl = _kernel_blocks(model, d, index=index, store=(rgb, dep))
            terms = [fw.masked_l1(s, g, m, 3) for (g, s, m, _) in wo.values()]
            terms.append(fw.masked_l1(fl, gt_flow, flow_mask, 2))
        loss = sum(w * t for w, t in zip((1.0, 0.7, 1.3, 0.01), terms[:-1] + [terms[-1]]))
        loss.backward()
        res[engine] = (wo, fl, [float(t) for t in terms], depth.grad.clone(), cam_l.grad.clone() if ba else None)
    wo_t, fl_t, terms_t, gd_t, gc_t = res["torch"]
    wo_h, fl_h, terms_h, gd_h, gc_h = res["hip"]
    # depth[0,5] < 0 puts the point behind (or near) the camera plane: pixel coordinates of size 1e4 .. 1e6, fp32 relative
    assert_close(fl_h, fl_t, 5e-3, 2e-5, "flow")
    for ps in patches:
        g_t, s_t, m_t, r_t = wo_t[ps]
        g_h, s_h, m_h, r_h = wo_h[ps]
        agree = (m_t == m_h).float().mean()
        assert agree > 0.9995, (ps, float(agree))
        assert torch.equal(g_t, g_h), ps
        both = (m_t & m_h)
        # a projected pixel coordinate of size ~1e3 carries a few fp32 ulps (1.2e-4 px each) of evaluation-order noise between
        # torch's matmuls and the kernel; on these white-noise frames (texel-to-texel differences ~1) that is up to ~5e-4 in value
        assert_close(s_h[both], s_t[both], 1e-3, 1e-4, f"sampled {ps}")
        if ps > 1:
            assert (r_t == r_h).float().mean() > 0.999
    for a, b in zip(terms_h, terms_t):
        assert abs(a - b) < 2e-4 * max(1.0, abs(b)), (terms_h, terms_t)
    # |sampled - gt| and the bilinear sample are continuous but only piecewise differentiable in the depth: a projected pixel
    # coordinate within the evaluation-order noise (~1e-4 px) of an integer picks the neighbouring texel cell in one of the two
    # implementations -- same value, different slope (on white-noise frames: O(1) different).  Expected: a handful of the 4800
    # (target, source, pixel) samples per patch cell; the depth gradients of those pixels are excluded, not loosened.
    tol = 2e-3 * float(gd_t.abs().max()) + 5e-3 * gd_t.abs()
    off = (gd_h - gd_t).abs() > tol
    print(f"[warp kinks] patches {patches} ba {ba}: {int(off.sum())} of {off.numel()} depth gradients outside tol; "
          f"max |dg| {float((gd_h - gd_t).abs().max()):.4g}, max |g_torch| {float(gd_t.abs().max()):.4g}, max |g_hip| {float(gd_h.abs().max()):.4g}")
    # counted and bounded (measured on MI355X, round 4: 0 or 1 of the 1200 depth gradients outside `tol`, largest difference 1.0e-5
    # where the largest gradient is 2.5e-3 .. 0.2): at most 6 kink pixels, and none of them differs by more than 2 % of the largest
    # gradient (the round-3 bound allowed 1 % of the pixels and 200 %)
    assert int(off.sum()) <= 6, int(off.sum())
    assert float((gd_h - gd_t).abs().max()) <= 2e-2 * float(gd_t.abs().max())
    if ba:
        assert_close(gc_h, gc_t, 2e-3 * float(gc_t.abs().max()), 5e-3, "d loss / d camera tensors")


def test_masked_l1_vs_torch_and_empty_selection():
    from nicer_slam_amd.fused.warp import masked_l1
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.rand(7, 33, 5, 3, device="cuda", generator=g).requires_grad_(True)
    b = torch.rand(7, 33, 5, 3, device="cuda", generator=g)
    b.view(-1)[::17] = a.detach().view(-1)[::17]                              # exact ties: sign(0) = 0
    m = torch.rand(7, 33, 5, device="cuda", generator=g) > 0.4
    loss = masked_l1(a, b, m, 3)
    (2.5 * loss).backward()
    a2 = a.detach().clone().requires_grad_(True)
    ref = (a2[m] - b[m]).abs().mean()
    (2.5 * ref).backward()
    assert abs(float(loss) - float(ref)) < 1e-6
    assert_close(a.grad, a2.grad, 1e-9, 1e-5, "masked L1 gradient")
    none = masked_l1(a, b, torch.zeros_like(m), 3)
    assert torch.isnan(none)                                                  # mean of an empty selection, like torch
    full = masked_l1(a, b, None, 3)
    assert abs(float(full) - float((a - b).abs().mean())) < 1e-6


def test_fused_mapping_with_flow_and_warp_under_bundle_adjustment_vs_reference():
    """End to end on the fused engine: reference golden (run A of reproj_case) -- flow, warp tensors, total camera gradient of
    rgb L1 + 0.5 (warp terms) + 0.1 flow L1 through renderer AND re-projection kernels, poses requiring grad."""
    from nicer_slam_amd.model.loss import SLAMLoss
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    from test_mapping_gpu import _run
    fx = load("reproj_blocks")
    edges = (tt(fx["in_idii"]).cuda(), tt(fx["in_idjj"]).cuda(), None, None)
    gt = {"full_rgb": tt(fx["in_full_rgb"]).cuda(), "full_depth": tt(fx["in_full_depth"]).cuda(), "edges": edges}
    crit = SLAMLoss("torch.nn.L1Loss", 0.0)
    model, cam, out = _run(fx, "fused", gt)
    assert model.last_engine == "fused"
    check_forward(fx, out["warp_output"], out["flow"], atol=5e-5)
    loss = (out["rgb_values"].reshape(-1, 3) - tt(fx["gt_rgb"]).cuda()).abs().mean()
    loss = loss + 0.5 * crit._warp_loss(out["warp_output"])
    loss = loss + 0.1 * crit.get_flow_loss(out, {"flow": tt(fx["gt_flow"]).cuda(), "flow_mask": tt(fx["gt_flow_mask"]).cuda()})
    loss.backward()
    assert abs(float(loss) - float(fx["out_loss"])) < 2e-5
    ref = fx["grad_cam"]
    assert_close(cam.grad, ref, 2e-3 * float(np.abs(ref).max()), 2e-3, "grad_cam")
