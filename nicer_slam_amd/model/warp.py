"""Patch-warp gather of the mapping objective (reference code/model/network.py:167-279, code/utils/general.py:129-145;
SURVEY 8f next-1).  Composed engine (torch ops): per-pixel work on a few thousand rays, not on the per-sample hot path.

For every patch size: the patch around each sampled pixel of every keyframe is lifted with the ray's RENDERED depth,
projected into every keyframe, and the full images are sampled there; the loss compares with the patch read directly
from the source image.  Returns {patchsize: (gt_warp_rgbs, target_sampled_rgb, total_warp_mask, depth_mask_ray_level)}
with the reference's shapes: [target, reference, n_pixels, patch^2, (3)]."""
import torch
import torch.nn.functional as F

from ..utils import rend_util


def uv2patch(uv, patchsize):
    """[b, n, 2] pixel centres -> [b, n, p, p, 2] pixel coordinates of the p x p patch (general.py:129-145)."""
    if patchsize == 1:
        return uv.clone().reshape(-1, uv.shape[1], 1, 1, 2)
    half = patchsize // 2
    r = torch.arange(-half, half + 1, device=uv.device)
    gx, gy = torch.meshgrid(r, r, indexing="ij")
    return uv.unsqueeze(2).unsqueeze(2) + torch.stack([gx, gy], -1).to(uv.dtype)[None, None]


def patch_warp(model, uv, pose, intrinsics, rendered_depth, ground_truth, batch_size):
    H, W = model.H, model.W
    stacked = lambda t: t.stacked() if hasattr(t, "stacked") else t       # fused/warp.py::FrameStore (feed.py) or a plain tensor
    full_rgb = stacked(ground_truth["full_rgb"]).reshape(batch_size, H, W, 3)
    full_depth = stacked(ground_truth["full_depth"]).reshape(batch_size, H, W, 1)
    depth = rendered_depth.reshape(batch_size, -1, 1, 1)          # z of the centre ray, shared by its patch
    w2c = torch.linalg.inv(pose)
    out = {}
    for ps in model.patchsizes:
        p2 = ps * ps
        uvp = uv2patch(uv, ps).reshape(batch_size, -1, 2)                              # [b, n*p2, 2]
        dirs, cam_loc = rend_util.get_camera_params(uvp, pose, intrinsics)
        pts = cam_loc[:, None, None, :] + depth * dirs.reshape(batch_size, -1, p2, 3)  # [b, n, p2, 3] world
        flat = pts.reshape(-1, 3).t()                                                   # [3, b*n*p2]
        cam = w2c[:, :3, :3] @ flat + w2c[:, :3, 3:]                                    # [b_t, 3, M]
        proj = (intrinsics[:, :3, :3] @ cam).permute(0, 2, 1).reshape(batch_size, batch_size, -1, p2, 3)
        tz = proj[..., 2:]
        tuv = proj[..., :2] / (tz + 1e-8)
        grid = torch.stack([tuv[..., 0] / W, tuv[..., 1] / H], -1) * 2 - 1.0           # [-1, 1], align_corners=True
        grid = grid.reshape(batch_size, -1, 1, 2)
        sampled = F.grid_sample(full_rgb.permute(0, 3, 1, 2), grid, mode="bilinear", padding_mode="zeros",
                                align_corners=True)
        sampled = sampled.reshape(batch_size, 3, batch_size, -1, p2).permute(0, 2, 3, 4, 1)
        tmask = ((grid[..., 0] > -1) & (grid[..., 0] < 1) & (grid[..., 1] > -1) & (grid[..., 1] < 1)
                 & (tz.reshape(batch_size, -1, 1) > 0)).reshape(batch_size, batch_size, -1, p2)
        # the same patches read straight from their own image (ones outside the image)
        inside = (uvp[..., 0] >= 0) & (uvp[..., 1] >= 0) & (uvp[..., 0] < W) & (uvp[..., 1] < H)   # [b, n*p2]
        ui = uvp[..., 0].long().clamp(0, W - 1)
        vi = uvp[..., 1].long().clamp(0, H - 1)
        bidx = torch.arange(batch_size, device=uv.device)[:, None].expand_as(ui)
        gt_rgb = torch.where(inside[..., None], full_rgb[bidx, vi, ui], torch.ones_like(full_rgb[bidx, vi, ui]))
        gt_dep = torch.where(inside[..., None], full_depth[bidx, vi, ui], torch.ones_like(full_depth[bidx, vi, ui]))
        gmask = inside[None].expand(batch_size, -1, -1).reshape(batch_size, batch_size, -1, p2)
        gt_rgbs = gt_rgb.reshape(1, batch_size, -1, p2, 3).repeat(batch_size, 1, 1, 1, 1)
        total = gmask & tmask
        ray_level = None
        if ps > 1:   # keep only patches whose ground-truth depth is locally flat (network.py:259-270)
            flat_ok = torch.var(gt_dep.reshape(batch_size, -1, p2), dim=-1, unbiased=False) < 0.01
            ray_level = flat_ok.reshape(-1)
            total = total & flat_ok[None, :, :, None].expand(batch_size, -1, -1, p2)
        out[ps] = (gt_rgbs, sampled, total, ray_level)
    return out


def flow_reproject(uv, pose, intrinsics, rendered_depth, edges):
    """Optical-flow reprojection (network.py:150-165): the rendered 3-D point of every pixel of frame idii[e], projected into
    frame idjj[e], minus the pixel.  [n_edges, n, 2]."""
    bs = uv.shape[0]
    dirs, cam_loc = rend_util.get_camera_params(uv, pose, intrinsics)
    pts = (cam_loc.unsqueeze(1) + rendered_depth.reshape(bs, -1, 1) * dirs).permute(0, 2, 1)        # [b, 3, n]
    idii, idjj = edges[0], edges[1]
    w2c = torch.linalg.inv(pose[idjj])
    cam_pts = w2c[:, :3, :3] @ pts[idii] + w2c[:, :3, 3:]
    proj = (intrinsics[idjj][:, :3, :3] @ cam_pts).permute(0, 2, 1)
    return proj[..., :2] / (proj[..., 2:] + 1e-8) - uv[idii]
