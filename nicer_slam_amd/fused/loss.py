"""The per-ray terms of SLAMLoss on the device in three launches (nsa_slam_loss, csrc/loss_terms.hip) instead of ~150 torch
launches plus their autograd graph -- and without the host round trips of the torch restatement (its empty-mask tests read device
scalars).  Reference: SLAMLoss.forward, code/model/loss.py:113-233; code/utils/MiDaS.py:6-143.

``fused_terms`` returns (weighted sum [differentiable], unweighted terms [7], detached): the Function computes the gradient of the
weighted sum with respect to rgb_values, depth_values, normal_map, grad_theta, grad_theta_nei in its forward (the weights are
Python floats known at call time) and scales it by the incoming cotangent in backward."""
import ctypes

import torch

from .._native import lib, check, LossDesc

TERMS = ("rgb", "eikonal", "smooth", "depth", "gt_depth", "normal_l1", "normal_cos")


def available(t):
    return t.is_cuda and t.dtype == torch.float32


class _FusedSlamLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, depth, normal, grad_theta, grad_theta_nei, rgb_gt, depth_mono, depth_real, depth_real_mask, mask_gt,
                sdf, normal_gt, weights, whole):
        dev = rgb.device
        bs, n = depth.shape[0], depth.shape[1]
        R = bs * n
        c = lambda t: t.detach().to(dev, torch.float32).contiguous()
        rgb_, depth_, normal_ = c(rgb).reshape(R, 3), c(depth).reshape(R), c(normal).reshape(R, 3)
        E = 0 if grad_theta is None else grad_theta.shape[0]
        gth = c(grad_theta) if E else None
        gnei = c(grad_theta_nei) if (E and grad_theta_nei is not None) else None
        keep = [rgb_, depth_, normal_, gth, gnei, c(rgb_gt).reshape(R, 3), c(depth_mono).reshape(R), c(depth_real).reshape(R),
                c(depth_real_mask).reshape(R), c(mask_gt).reshape(R), c(sdf).reshape(R, -1), c(normal_gt).reshape(R, 3)]
        S = keep[10].shape[1]
        # the five gradients live in ONE buffer: one launch scales them by the incoming cotangent in backward, and the two eikonal
        # halves stay adjacent, as fused/mapping.py::FusedSdfGradient.backward wants them (no cat of the halves)
        sizes = [R * 3, R, R * 3, E * 3 if E else 0, E * 3 if gnei is not None else 0]
        gbuf = torch.empty(sum(sizes), device=dev)
        o = [0]
        for n_ in sizes:
            o.append(o[-1] + n_)
        g_rgb, g_depth, g_normal = gbuf[o[0]:o[1]].view(R, 3), gbuf[o[1]:o[2]], gbuf[o[2]:o[3]].view(R, 3)
        g_theta = gbuf[o[3]:o[4]].view(E, 3) if E else None
        g_nei = gbuf[o[4]:o[5]].view(E, 3) if gnei is not None else None
        terms = torch.empty(8, device=dev)
        ws = torch.empty((int(lib.nsa_slam_loss_workspace(bs, n, E)) + 1) // 2, device=dev, dtype=torch.float64)
        p = lambda t: None if t is None else t.data_ptr()
        d = LossDesc(bs, n, S, E, p(rgb_), p(keep[5]), p(depth_), p(keep[6]), p(keep[7]), p(keep[8]), p(keep[9]), p(keep[10]),
                     p(normal_), p(keep[11]), p(gth), p(gnei), *[float(w) for w in weights], int(bool(whole)),
                     p(g_rgb), p(g_depth), p(g_normal), p(g_theta), p(g_nei), p(terms))
        check(lib.nsa_slam_loss(ctypes.byref(d), ws.data_ptr(), torch.cuda.current_stream().cuda_stream))
        ctx.shapes = (rgb.shape, depth.shape, normal.shape, (E, 3))
        ctx.save_for_backward(gbuf)
        ctx.offsets, ctx.have = o, (g_theta is not None, g_nei is not None)
        ctx.mark_non_differentiable(terms)
        return terms[7].clone(), terms

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        (gbuf,) = ctx.saved_tensors
        g = gbuf * g_total
        o = ctx.offsets
        s_rgb, s_depth, s_normal, s_theta = ctx.shapes
        return (g[o[0]:o[1]].view(s_rgb), g[o[1]:o[2]].view(s_depth), g[o[2]:o[3]].view(s_normal),
                g[o[3]:o[4]].view(s_theta) if ctx.have[0] else None, g[o[4]:o[5]].view(s_theta) if ctx.have[1] else None) + (None,) * 9


def fused_terms(model_outputs, rgb_gt, depth_mono, depth_real, depth_real_mask, mask_gt, normal_gt, weights, whole=False,
                use_eikonal=True, use_smooth=True):
    """weights = (rgb, eikonal, smooth, depth, gt_depth, normal_l1, normal_cos).  -> (weighted sum, terms[8])"""
    gt = model_outputs.get("grad_theta") if (use_eikonal or use_smooth) else None
    gn = model_outputs.get("grad_theta_nei") if (use_smooth and gt is not None) else None
    return _FusedSlamLoss.apply(model_outputs["rgb_values"], model_outputs["depth_values"], model_outputs["normal_map"], gt, gn,
                                rgb_gt, depth_mono, depth_real, depth_real_mask, mask_gt, model_outputs["sdf"], normal_gt,
                                tuple(weights), whole)
