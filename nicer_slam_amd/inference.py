"""Batch inference consumers of the render core: full-image rendering in pixel chunks and SDF
evaluation on dense grids for mesh extraction.  Same kernels as the training path, no gradients, large batches.

Reference: utils.general.split_input / merge_output (code/utils/general.py:169-204), the vis loop of
VolSDFTrainRunner.vis (code/training/volsdf_train.py:255-290), get_grid_uniform / get_surface_trace's grid evaluation
(code/utils/plots.py:102-166).  Marching cubes / PNG / PLY writing stay with the caller (offline tooling, out of scope).
"""
import ctypes

import torch

from ._native import lib, check
from .fused.sampler import grid_desc, packed_sdf, precision_of, sdf_grid_desc, supported as fused_supported


def split_input(model_input, total_pixels, n_pixels=10000):
    """List of per-chunk copies of ``model_input`` (uv and the optional per-pixel entries sliced along dim 1)."""
    out = []
    dev = model_input["uv"].device
    for idx in torch.split(torch.arange(total_pixels, device=dev), n_pixels, dim=0):
        data = dict(model_input)
        for key in ("uv", "object_mask", "depth", "gt_depth"):
            if key in data:
                data[key] = torch.index_select(model_input[key], 1, idx.to(model_input[key].device))
        out.append(data)
    return out


def merge_output(res, total_pixels, batch_size):
    """Concatenate per-chunk output dicts back to [batch*total_pixels(, C)] like the reference."""
    merged = {}
    for key, first in res[0].items():
        if first is None:
            continue
        if first.dim() == 1:
            merged[key] = torch.cat([r[key].reshape(batch_size, -1, 1) for r in res], 1).reshape(batch_size * total_pixels)
        else:
            merged[key] = torch.cat([r[key].reshape(batch_size, -1, r[key].shape[-1]) for r in res], 1).reshape(
                batch_size * total_pixels, -1)
    return merged


@torch.no_grad()
def render_image(model, model_input, indices=None, ground_truth=None, mode="tracking_vis", n_pixels=65536,
                 stage="fine", color_stage="highfreq"):
    """Render every pixel of ``model_input['uv']`` ([b, H*W, 2]) in chunks of ``n_pixels`` rays; returns the merged
    ``rgb_values``, ``normal_map``, ``depth_values``.  Call on a model in eval mode (deterministic sampler)."""
    total = model_input["uv"].shape[1]
    bs = model_input["uv"].shape[0]
    if indices is None:
        indices = torch.arange(bs, device=model_input["uv"].device)
    res = []
    for chunk in split_input(model_input, total, n_pixels):
        out = model(chunk, indices, ground_truth or {}, mode=mode, stage=stage, color_stage=color_stage)
        res.append({k: out[k].detach() for k in ("rgb_values", "normal_map", "depth_values")})
    return merge_output(res, total, bs)


def get_grid_uniform(resolution, grid_boundary=(-2.0, 2.0), device="cpu"):
    """Axis values and the [res^3, 3] point list in the reference's order (np.meshgrid 'xy' indexing: the flat index
    runs over (y, x, z))."""
    x = torch.linspace(grid_boundary[0], grid_boundary[1], resolution, dtype=torch.float64, device=device)
    yy, xx, zz = torch.meshgrid(x, x, x, indexing="ij")
    pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1).float()
    return {"grid_points": pts, "shortest_axis_length": 2.0, "xyz": [x, x, x], "shortest_axis_index": 0}


@torch.no_grad()
def sdf_values(model, points, stage="fine", chunk=1 << 22):
    """SDF (coarse + fine) at ``points`` [N,3] on the GPU, no gradients: ImplicitNetworkGrid_COMBINE.get_sdf_vals[:, 0]."""
    if not (points.is_cuda and fused_supported(model)):
        raise RuntimeError("sdf_values: needs CUDA points and a model configuration covered by the fused kernels")
    imp = model.implicit_network
    gc, keep_c = sdf_grid_desc(model, "coarse", "sampler")
    gf, keep_f = sdf_grid_desc(model, "fine", "sampler")
    pc, pf = packed_sdf(model, "coarse", use="sampler"), packed_sdf(model, "fine", use="sampler")
    points = points.contiguous().float()
    out = torch.empty(points.shape[0], device=points.device)
    st = torch.cuda.current_stream().cuda_stream
    fine = stage != "coarse"
    for lo in range(0, points.shape[0], chunk):
        n = min(chunk, points.shape[0] - lo)
        check(lib.nsa_sdf_points(points[lo:lo + n].data_ptr(), n, ctypes.byref(gc), ctypes.byref(gf) if fine else None,
                                 pc.data_ptr(), pf.data_ptr() if fine else None, out[lo:lo + n].data_ptr(), st))
    return out


@torch.no_grad()
def sdf_grid(model, resolution, grid_boundary=(-2.0, 2.0), stage="fine", chunk=1 << 22):
    """SDF volume [res, res, res] indexed (x, y, z) -- what get_surface_trace hands to marching cubes (plots.py:
    121-127: reshape(ny, nx, nz).transpose(1, 0, 2)) -- evaluated chunk by chunk without materialising the point list."""
    dev = model.voxels.device
    ax = torch.linspace(grid_boundary[0], grid_boundary[1], resolution, dtype=torch.float64, device=dev).float()
    n = resolution ** 3
    out = torch.empty(n, device=dev)
    for lo in range(0, n, chunk):
        flat = torch.arange(lo, min(lo + chunk, n), device=dev)
        iy = flat // (resolution * resolution)
        ix = (flat // resolution) % resolution
        iz = flat % resolution
        pts = torch.stack([ax[ix], ax[iy], ax[iz]], -1)
        out[lo:lo + flat.numel()] = sdf_values(model, pts, stage, chunk)
    return out.view(resolution, resolution, resolution).permute(1, 0, 2).contiguous()
