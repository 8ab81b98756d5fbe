"""Keyframe re-projection blocks of a mapping iteration on the device (C ABI section 5, csrc/warp_terms.hip): patch warp,
flow and their masked-L1 loss terms as autograd Functions over the rendered depth (and, under bundle adjustment, the poses).

Reference: code/model/network.py:153-165 (flow), :167-279 (patch warp), code/utils/general.py:129-145 (uv2patch),
code/model/loss.py:106-111,136-142 (the L1 terms).  The torch restatement of the same blocks is model/warp.py (composed engine).
"""
import ctypes

import torch

from .._native import lib, check, WarpDesc
from ..hashencoder.backend import _timed


def _inv(pose):
    """torch.inverse of the [b,4,4] poses (code/model/network.py:156-157, 190-191) without torch.linalg.inv's error check: that check
    reads the LU status back (`info.any().item()`), a device synchronisation in the middle of every mapping forward.  Same factorisation,
    same values and gradient; a singular pose yields inf / nan instead of raising."""
    return torch.linalg.inv_ex(pose).inverse


def _stream():
    return torch.cuda.current_stream().cuda_stream


def available(depth, uv, pose):
    return depth.is_cuda and depth.dtype == torch.float32 and uv.dtype == torch.float32 and pose.dtype == torch.float32


class FrameStore:
    """Resident full frames + the batch's frame indices: what ground_truth['full_rgb'] / ['full_depth'] carry when the feed
    keeps its frames in one store (feed.py) -- the kernels index the store, nothing is stacked per iteration."""

    def __init__(self, images, index):
        self.images = images        # [frames, H*W, C] (or [frames, H, W, C]) fp32, contiguous
        self.index = index          # [b] int32 device tensor

    def stacked(self):
        """The reference's [b, H*W, C] tensor (for the composed engine / reference-shaped consumers)."""
        return self.images.index_select(0, self.index.long())


def _frames(t):
    if isinstance(t, FrameStore):
        return t.images, t.index
    return t, None


def _desc(uv, pose, w2c, K, depth, H, W, images=None, depths=None, index=None):
    b, n = uv.shape[0], uv.shape[1]
    p = lambda t: None if t is None else t.data_ptr()
    return WarpDesc(b, n, H, W, p(uv), p(pose), p(w2c), p(K), p(depth), p(images), p(depths), p(index))


def _c(t):
    return t.detach().to(torch.float32).contiguous()


class _PatchWarp(torch.autograd.Function):
    """(depth[b,n], pose[b,4,4], w2c[b,4,4]) -> sampled[b,b,n,p2,3]; also returns mask, gt_rgb, flat (non-differentiable)."""

    @staticmethod
    def forward(ctx, depth, pose, w2c, uv, K, images, depths, index, H, W, patch):
        b, n = uv.shape[0], uv.shape[1]
        p2 = patch * patch
        dev = depth.device
        depth_, pose_, w2c_, uv_, K_ = _c(depth).reshape(b, n), _c(pose), _c(w2c), _c(uv), _c(K)
        sampled = torch.empty(b, b, n, p2, 3, device=dev)
        gt_rgb = torch.empty(b, b, n, p2, 3, device=dev)
        mask = torch.empty(b, b, n, p2, device=dev, dtype=torch.bool)
        flat = torch.empty(b, n, device=dev, dtype=torch.bool) if patch > 1 else None
        d = _desc(uv_, pose_, w2c_, K_, depth_, H, W, images, depths, index)
        with _timed("k_warp_fwd", sampled.numel() * 4):
            check(lib.nsa_patch_warp_forward(ctypes.byref(d), patch, sampled.data_ptr(), mask.data_ptr(), gt_rgb.data_ptr(),
                                             None if flat is None else flat.data_ptr(), _stream()))
        ctx.save_for_backward(depth_, pose_, w2c_, uv_, K_)
        ctx.frames = (images, depths, index)
        ctx.meta = (H, W, patch, depth.shape)
        ctx.set_materialize_grads(False)
        if flat is None:
            ctx.mark_non_differentiable(mask, gt_rgb)
            return sampled, mask, gt_rgb
        ctx.mark_non_differentiable(mask, gt_rgb, flat)          # ONE call: a second call replaces the first one's list
        return sampled, mask, gt_rgb, flat

    @staticmethod
    def backward(ctx, g_sampled, *_):
        if g_sampled is None:                                     # (set_materialize_grads(False): the samples were not used)
            return (None,) * 11
        depth_, pose_, w2c_, uv_, K_ = ctx.saved_tensors
        images, depths, index = ctx.frames
        H, W, patch, dshape = ctx.meta
        b, n = uv_.shape[0], uv_.shape[1]
        dev = depth_.device
        want_pose = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        g_depth = torch.empty(b, n, device=dev)
        g_pose = torch.empty(b, 4, 4, device=dev) if want_pose else None
        g_w2c = torch.empty(b, 4, 4, device=dev) if want_pose else None
        nws = int(lib.nsa_patch_warp_workspace(b, n, patch, int(want_pose)))
        ws = torch.empty(nws, device=dev) if nws else None
        d = _desc(uv_, pose_, w2c_, K_, depth_, H, W, images, depths, index)
        p = lambda t: None if t is None else t.data_ptr()
        with _timed("k_warp_bwd", g_sampled.numel() * 4):
            check(lib.nsa_patch_warp_backward(ctypes.byref(d), patch, g_sampled.contiguous().data_ptr(), g_depth.data_ptr(),
                                              p(g_pose), p(g_w2c), p(ws), _stream()))
        return (g_depth.reshape(dshape), g_pose if ctx.needs_input_grad[1] else None,
                g_w2c if ctx.needs_input_grad[2] else None) + (None,) * 8


class _Flow(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, pose, w2c, uv, K, idii, idjj, H, W):
        b, n = uv.shape[0], uv.shape[1]
        ne = idii.shape[0]
        dev = depth.device
        depth_, pose_, w2c_, uv_, K_ = _c(depth).reshape(b, n), _c(pose), _c(w2c), _c(uv), _c(K)
        ii, jj = idii.to(dev, torch.int64).contiguous(), idjj.to(dev, torch.int64).contiguous()
        flow = torch.empty(ne, n, 2, device=dev)
        d = _desc(uv_, pose_, w2c_, K_, depth_, H, W)
        check(lib.nsa_flow_forward(ctypes.byref(d), ii.data_ptr(), jj.data_ptr(), ne, flow.data_ptr(), _stream()))
        ctx.save_for_backward(depth_, pose_, w2c_, uv_, K_, ii, jj)
        ctx.meta = (H, W, depth.shape)
        return flow

    @staticmethod
    def backward(ctx, g_flow):
        depth_, pose_, w2c_, uv_, K_, ii, jj = ctx.saved_tensors
        H, W, dshape = ctx.meta
        b, n = uv_.shape[0], uv_.shape[1]
        ne = ii.shape[0]
        dev = depth_.device
        want_pose = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        g_depth = torch.empty(b, n, device=dev)
        g_pose = torch.empty(b, 4, 4, device=dev) if want_pose else None
        g_w2c = torch.empty(b, 4, 4, device=dev) if want_pose else None
        nws = int(lib.nsa_flow_workspace(b, n, ne, int(want_pose)))
        ws = torch.empty(nws, device=dev) if nws else None
        d = _desc(uv_, pose_, w2c_, K_, depth_, H, W)
        p = lambda t: None if t is None else t.data_ptr()
        check(lib.nsa_flow_backward(ctypes.byref(d), ii.data_ptr(), jj.data_ptr(), ne, g_flow.contiguous().data_ptr(),
                                    g_depth.data_ptr(), p(g_pose), p(g_w2c), p(ws), _stream()))
        return (g_depth.reshape(dshape), g_pose if ctx.needs_input_grad[1] else None,
                g_w2c if ctx.needs_input_grad[2] else None) + (None,) * 6


class _MaskedL1(torch.autograd.Function):
    """mean |pred[mask] - target[mask]| with its gradient formed in the forward (three launches, no host round trip)."""

    @staticmethod
    def forward(ctx, pred, target, mask, channels):
        dev = pred.device
        pred_, target_ = _c(pred), _c(target).to(dev)
        items = pred_.numel() // channels
        # the kernel indexes target like pred and mask per item: anything else (a broadcast target, a mask with a trailing
        # dimension) would read out of bounds where the reference's boolean indexing raises (loss.py:106-111)
        if pred_.numel() != items * channels or pred.shape[-1] != channels:
            raise ValueError(f"masked_l1: pred {tuple(pred.shape)} is not [..., {channels}]")
        if target_.shape != pred_.shape:
            raise ValueError(f"masked_l1: target {tuple(target.shape)} must have the shape of pred {tuple(pred.shape)}")
        m = None
        if mask is not None:
            if mask.numel() != items:
                raise ValueError(f"masked_l1: mask {tuple(mask.shape)} must select among the {items} items of pred {tuple(pred.shape)}")
            m = mask.to(dev).contiguous()
            m = m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
        loss = torch.empty(1, device=dev)
        g = torch.empty_like(pred_) if ctx.needs_input_grad[0] else None
        ws = torch.empty((int(lib.nsa_masked_l1_workspace(items)) + 1) // 2, device=dev, dtype=torch.float64)
        check(lib.nsa_masked_l1(pred_.data_ptr(), target_.data_ptr(), None if m is None else m.data_ptr(), items, channels,
                                loss.data_ptr(), None if g is None else g.data_ptr(), ws.data_ptr(), _stream()))
        ctx.g = g
        ctx.shape = pred.shape
        return loss[0]

    @staticmethod
    def backward(ctx, g_loss):
        g = ctx.g
        return (None if g is None else (g * g_loss).reshape(ctx.shape)), None, None, None


def masked_l1(pred, target, mask, channels):
    """pred, target: [..., channels]; mask: [...] bool or None."""
    return _MaskedL1.apply(pred, target, mask, channels)


def patch_warp(model, uv, pose, intrinsics, rendered_depth, ground_truth, batch_size):
    """Same contract as model/warp.py::patch_warp: {patch: (gt_warp_rgbs, target_sampled_rgb, total_warp_mask, ray_level)}."""
    H, W = model.H, model.W
    images, index = _frames(ground_truth["full_rgb"])
    depths, _ = _frames(ground_truth["full_depth"]) if "full_depth" in ground_truth else (None, None)
    images = images.to(uv.device, torch.float32).contiguous()
    depths = None if depths is None else depths.to(uv.device, torch.float32).contiguous()
    w2c = _inv(pose)                                                 # network.py:190-191
    K = intrinsics.to(uv.device)
    out = {}
    for ps in model.patchsizes:
        ps = int(ps)
        res = _PatchWarp.apply(rendered_depth, pose, w2c, uv, K, images, depths, index, H, W, ps)
        sampled, mask, gt_rgb = res[:3]
        out[ps] = (gt_rgb, sampled, mask, res[3].reshape(-1) if ps > 1 else None)
    return out


def flow(model, uv, pose, intrinsics, rendered_depth, edges):
    idii, idjj = edges[0], edges[1]
    w2c = _inv(pose)                                                 # network.py:156-157 (inverse of every pose, gathered in-kernel)
    return _Flow.apply(rendered_depth, pose, w2c, uv, intrinsics.to(uv.device), idii, idjj, model.H, model.W)
