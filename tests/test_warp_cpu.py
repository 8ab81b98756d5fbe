"""The torch restatement of the keyframe re-projection blocks (model/warp.py: flow + patch warp; composed engine) against the
function-level golden captured from the reference's SLAMNetwork.forward (tests/golden/make_golden.py::reproj_case,
code/model/network.py:153-279): forward tensors, d/d(rendered depth) and the direct pose gradient of every masked-L1 term."""
from types import SimpleNamespace

import numpy as np
import torch

from helpers import assert_close, load, tt


def reproj_inputs(fx, device="cpu"):
    H, W = [int(v) for v in fx["meta_img_res"]]
    d = dict(uv=tt(fx["in_uv"]).to(device), K=tt(fx["in_K"]).to(device),
             pose=tt(fx["in_pose"]).to(device).requires_grad_(True),
             depth=tt(fx["in_rendered_depth"]).to(device).requires_grad_(True),
             full_rgb=tt(fx["in_full_rgb"]).to(device), full_depth=tt(fx["in_full_depth"]).to(device),
             edges=(tt(fx["in_idii"]).to(device), tt(fx["in_idjj"]).to(device), None, None),
             gt_flow=tt(fx["gt_flow"]).to(device), flow_mask=tt(fx["gt_flow_mask"]).to(device))
    return d, SimpleNamespace(H=H, W=W, patchsizes=[1, 5])


def check_forward(fx, warp_out, flow, atol=2e-5):
    assert_close(flow, fx["out_flow"], 2e-4, 1e-5, "flow (pixels)")
    for ps, (gt_w, samp, mask, ray) in warp_out.items():
        ref_mask = tt(fx[f"out_warp{ps}_mask"])
        same = (mask.cpu() == ref_mask)
        assert same.float().mean() > 0.995, (ps, same.float().mean())     # a projection within an ulp of the image border may flip
        assert_close(gt_w, fx[f"out_warp{ps}_gt"], 0, 0, f"warp{ps} gt")
        assert_close(samp, fx[f"out_warp{ps}_sampled"], atol, 1e-5, f"warp{ps} sampled")
        if ps > 1:
            assert torch.equal(ray.cpu(), tt(fx[f"out_warp{ps}_raymask"]))


def check_backward(fx, terms, depth, pose, gtol=1e-3):
    """terms: {tag: scalar}; one backward per term like the golden."""
    for tag, term in terms.items():
        depth.grad = None
        pose.grad = None
        term.backward(retain_graph=True)
        for name, got in (("depth", depth.grad), ("pose", pose.grad)):
            ref = fx[f"grad_{name}_{tag}"]
            assert_close(got.reshape(ref.shape), ref, 1e-7 + gtol * float(np.abs(ref).max()), gtol, f"d {tag} / d {name}")


def test_torch_twin_matches_reference_reprojection_blocks():
    from nicer_slam_amd.model.warp import flow_reproject, patch_warp
    fx = load("reproj_blocks")
    d, model = reproj_inputs(fx)
    bs = d["uv"].shape[0]
    warp_out = patch_warp(model, d["uv"], d["pose"], d["K"], d["depth"].reshape(-1, 1).unsqueeze(2),
                          {"full_rgb": d["full_rgb"], "full_depth": d["full_depth"]}, bs)
    flow = flow_reproject(d["uv"], d["pose"], d["K"], d["depth"], d["edges"])
    check_forward(fx, warp_out, flow)
    terms = {f"warp{ps}": (s[m] - g[m]).abs().mean() for ps, (g, s, m, _) in warp_out.items()}
    terms["flow"] = (flow[d["flow_mask"]] - d["gt_flow"][d["flow_mask"]]).abs().mean()
    for (tag, t), ref in zip(terms.items(), list(fx["out_warp_terms"]) + [fx["out_flow_term"]]):
        assert abs(float(t) - float(ref)) < 1e-5 * max(1.0, abs(float(ref))), tag
    check_backward(fx, terms, d["depth"], d["pose"])
