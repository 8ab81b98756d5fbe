"""Fused composite pass: per-point SDF networks + colour network + per-ray compositing and their hand-derived
backward, as one torch.autograd.Function over (rays_o, rays_d) -- thin glue over Section 2 of the C ABI.

Replaces, for the data path (pose gradient; what a tracking iteration needs), the body of SLAMNetwork.forward
between the sampler and the output dict (reference code/model/network.py:112-151, 338-345).  Parameter gradients
(mapping) still go through the composed engine.
"""
import ctypes
import os

import torch

from .._native import lib, check, PointsDesc
from ..hashencoder.backend import _timed
from . import pack
from .sampler import COLOUR_COARSE_BWD, COLOUR_FWD_TRACK, forward_pair_ok, grid_desc, packed_sdf, precision_of, sdf_grid_desc, tile_of


def hl_size(P):
    return ((P + 31) // 32) * 2048


def save_size(P):
    """floats of the colour forward's save area: per 32-point tile 4096 (16 features + 48 Jacobian entries per lane) + 256 (ReLU masks of
    both hidden layers and the sigmoid outputs: what the data-path backward needs instead of recomputing the MLP; csrc/render_colour.hip)"""
    return ((P + 31) // 32) * (4096 + 256)


def hl_index(P, device):
    """[P,64] gather index into an HL buffer (tests / debugging)."""
    pid = torch.arange(P, device=device).unsqueeze(1)
    f = torch.arange(64, device=device).unsqueeze(0)
    tile, p = pid // 32, pid % 32
    t, rho = f // 32, f % 32
    h = (rho >> 2) & 1
    r = (rho & 3) + 4 * (rho >> 3)
    return (tile * 32 + 16 * t + r) * 64 + p + 32 * h


def packed_colour(model):
    net = model.rendering_network
    key = tuple((p.data_ptr(), p._version) for p in net.mlp_parameters())
    cache = model.__dict__.setdefault("_fused_pack", {})
    hit = cache.get("colour")
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():
        packed = pack.pack_colour_net(net)
    cache["colour"] = (key, packed)
    return packed


def supported(model):
    from .sampler import supported as sampler_ok
    rn = model.rendering_network
    return (sampler_ok(model) and rn.mode == "idr" and rn.use_grid_feature and not rn.per_image_code
            and rn.multires_view == 4 and rn.dims == [129, 64, 64, 3] and rn.encoding.num_levels == 16
            and rn.encoding.level_dim == 2 and not model.white_bkgd)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _pts(rays_o, rays_d, z_vals, order=None):
    R, S = z_vals.shape
    return PointsDesc(rays_o.data_ptr(), rays_d.data_ptr(), z_vals.data_ptr(), None, R * S, S,
                      None if order is None else order.data_ptr())


def _table_grad(param):
    """Where a MAP kernel adds a table's gradient: the table's persistent in-place buffer (fused/tablegrad.py) or, with
    NSA_TABLE_GRADS=autograd, a fresh zero-filled tensor handed back through autograd."""
    from . import tablegrad
    return tablegrad.target(param) if tablegrad.IN_PLACE else torch.zeros_like(param)


def _table_result(gt):
    from . import tablegrad
    return None if tablegrad.IN_PLACE else gt


# bits of the 30-bit Morton code the launch order is sorted on: 24 = a 256^3 lattice, three radix passes; the points of one such
# cell keep their ray order.  (Interleaved A/B of a mapping iteration, tools/ab_mapping.py: 23-25 bits 0.05-0.15 ms faster than the
# full 30 -- one pass less, and the MAP kernels do not lose by it; profiles/r05_ab_experiments.txt r5y/r5z.)
MORTON_BITS = 24


def morton_order(pts_desc, P, device):
    """int32 [P] spatially coherent launch order of the points described by ``pts_desc`` (argsort of Morton keys)."""
    order = torch.empty(P, device=device, dtype=torch.int32)
    if P == 0:
        return order
    ws = torch.empty(int(lib.nsa_morton_order_workspace(P)), device=device, dtype=torch.int32)
    check(lib.nsa_morton_order(ctypes.byref(pts_desc), order.data_ptr(), ws.data_ptr(), MORTON_BITS, _stream()))
    return order


def composite_forward_raw(model, rays_o, rays_d, z_vals, stage, need_bwd, sort_points=False, composite=True, track=None):
    """Launch the forward kernels of the composite pass.  Returns a dict of device buffers.
    ``composite=False`` stops after the per-point kernels (the tracker's nsa_composite_track forms the ray sums itself).
    ``track`` = dict(gt [R,3], ray_loss [R]) with ``composite=False``: when a ray is one workgroup of the colour forward (128 samples
    per ray) that launch also runs the tracking objective's composite + L1 + composite backward (nsa_colour_forward_track); the
    cotangents wait in b["track_out"] for composite_backward_raw(track=...), which then skips nsa_composite_track.
    ``sort_points``: run the per-point kernels in Morton order (mapping: the table-gradient scatter merges far more
    rows and the colour-table gathers share cache lines; the sort costs more than it saves for a 1024-ray tracking step)."""
    R, S = z_vals.shape
    P = R * S
    dev = z_vals.device
    imp = model.implicit_network
    gc, keep_c = sdf_grid_desc(model, "coarse")
    gf, keep_f = sdf_grid_desc(model, "fine")
    gr, keep_r = grid_desc(model.rendering_network.encoding, model.rendering_network.divide_factor, 2, precision_of(model, "colour"))
    pc, pf, pr = packed_sdf(model, "coarse"), packed_sdf(model, "fine"), packed_colour(model)
    order = morton_order(_pts(rays_o, rays_d, z_vals), P, dev) if sort_points else None
    pts = _pts(rays_o, rays_d, z_vals, order)
    b = dict(order=order, sdf=torch.empty(P, device=dev), grad=torch.empty(P, 3, device=dev), feat=torch.empty(hl_size(P), device=dev),
             rgb=torch.empty(P, 3, device=dev), save=torch.empty(save_size(P), device=dev) if need_bwd else None,
             weights=torch.empty(R, S, device=dev), rgb_values=torch.empty(R, 3, device=dev),
             depth=torch.empty(R, device=dev), nmap=torch.empty(R, 3, device=dev), entropy=torch.empty(R, device=dev),
             vox=model.voxels.contiguous(), packs=(pc, pf, pr), keep=(keep_c, keep_f, keep_r))
    st = _stream()
    if stage != "coarse" and forward_pair_ok(model):          # ImplicitNetworkGrid_COMBINE in one launch (quad tiling)
        gcp, keep_cp = sdf_grid_desc(model, "coarse", "coarse_pair")
        pcp = packed_sdf(model, "coarse", use="coarse_pair")
        b["keep"] += (keep_cp, pcp)
        with _timed("k_sdfnet_fwd<pair>", P * (4 * 8 * 8 * 4 + 8 * 8 * 4 * 4)):
            check(lib.nsa_sdfnet_forward_pair(ctypes.byref(pts), ctypes.byref(gcp), ctypes.byref(gf), pcp.data_ptr(), pf.data_ptr(),
                                              b["sdf"].data_ptr(), b["grad"].data_ptr(), b["feat"].data_ptr(), st))
    else:
        with _timed("k_sdfnet_fwd<coarse>", P * 4 * 8 * 8 * 4):
            check(lib.nsa_sdfnet_forward(ctypes.byref(pts), ctypes.byref(gc), pc.data_ptr(), 0, b["sdf"].data_ptr(),
                                         b["grad"].data_ptr(), b["feat"].data_ptr(), st))
        if stage != "coarse":
            with _timed("k_sdfnet_fwd<fine>", P * 8 * 8 * 4 * 4):
                check(lib.nsa_sdfnet_forward(ctypes.byref(pts), ctypes.byref(gf), pf.data_ptr(), 1, b["sdf"].data_ptr(),
                                             b["grad"].data_ptr(), b["feat"].data_ptr(), st))
    if track is not None and not composite and COLOUR_FWD_TRACK and S == 128 and order is None:
        t_out = dict(g_sdf=torch.empty(P, device=dev), g_rgb=torch.empty(P, 3, device=dev), g_grad=torch.empty(P, 3, device=dev))
        with _timed("k_colour_fwd", P * 16 * 8 * 2 * 4):
            check(lib.nsa_colour_forward_track(ctypes.byref(pts), ctypes.byref(gr), pr.data_ptr(), b["grad"].data_ptr(),
                                               b["feat"].data_ptr(), b["rgb"].data_ptr(), b["save"].data_ptr() if need_bwd else None,
                                               b["sdf"].data_ptr(), b["vox"].data_ptr(), model.voxel_res, track["gt"].data_ptr(), R,
                                               b["rgb_values"].data_ptr(), track["ray_loss"].data_ptr(), t_out["g_sdf"].data_ptr(),
                                               t_out["g_rgb"].data_ptr(), t_out["g_grad"].data_ptr(), st))
        b["track_out"] = t_out
        return b
    if composite and COLOUR_FWD_TRACK and S == 128 and order is None:      # the composite forward rides in the colour forward's launch
        with _timed("k_colour_fwd", P * 16 * 8 * 2 * 4):
            check(lib.nsa_colour_forward_composite(ctypes.byref(pts), ctypes.byref(gr), pr.data_ptr(), b["grad"].data_ptr(),
                                                   b["feat"].data_ptr(), b["rgb"].data_ptr(), b["save"].data_ptr() if need_bwd else None,
                                                   b["sdf"].data_ptr(), b["vox"].data_ptr(), model.voxel_res, b["weights"].data_ptr(),
                                                   b["rgb_values"].data_ptr(), b["depth"].data_ptr(), b["nmap"].data_ptr(),
                                                   b["entropy"].data_ptr(), st))
        return b
    with _timed("k_colour_fwd", P * 16 * 8 * 2 * 4):
        check(lib.nsa_colour_forward(ctypes.byref(pts), ctypes.byref(gr), pr.data_ptr(), b["grad"].data_ptr(),
                                     b["feat"].data_ptr(), b["rgb"].data_ptr(),
                                     b["save"].data_ptr() if need_bwd else None, st))
    if not composite:
        return b
    with _timed("k_composite_fwd", P * 32):
        check(lib.nsa_composite_forward(rays_o.data_ptr(), rays_d.data_ptr(), z_vals.data_ptr(), b["sdf"].data_ptr(),
                                        b["rgb"].data_ptr(), b["grad"].data_ptr(), b["vox"].data_ptr(), model.voxel_res,
                                        R, S, b["weights"].data_ptr(), b["rgb_values"].data_ptr(), b["depth"].data_ptr(),
                                        b["nmap"].data_ptr(), b["entropy"].data_ptr(), st))
    return b


def composite_backward_raw(model, rays_o, rays_d, z_vals, b, stage, color_stage, g_rgbv=None, g_depth=None, g_nmap=None,
                           g_ent=None, g_w=None, params=None, track=None, reduce_rays=True):
    """Launch the backward kernels.  Returns (g_rays_o[R,3], g_rays_d[R,3]); with ``params`` (dict of wanted parameter
    gradients: flat_c, flat_r, tab_c, tab_f, tab_r -- see fused/mapping.py) the MAP kernels run instead and a third
    value, the dict of those gradients, is returned.
    ``track`` = dict(gt [R,3], ray_loss [R]): the tracking objective -- nsa_composite_track forms the ray colours, the L1 cotangent
    and the composite backward in one launch (after composite_forward_raw(composite=False)).  ``reduce_rays=False`` returns the
    per-sample (g_x, g_dir) instead of the ray sums (the tracker's nsa_track_finish adds them)."""
    R, S = z_vals.shape
    P = R * S
    dev = z_vals.device
    imp = model.implicit_network
    gc, keep_c = sdf_grid_desc(model, "coarse")
    gf, keep_f = sdf_grid_desc(model, "fine")
    gr, keep_r = grid_desc(model.rendering_network.encoding, model.rendering_network.divide_factor, 2, precision_of(model, "colour"))
    pc, pf, pr = b["packs"]
    order = b.get("order")
    pts = _pts(rays_o, rays_d, z_vals, order)
    st = _stream()
    ptr = lambda t: None if t is None else t.data_ptr()
    gs = [None if g is None else g.contiguous() for g in (g_rgbv, g_depth, g_nmap, g_ent, g_w)]
    done = b.get("track_out") if track is not None else None       # the colour forward's launch already ran the tracking objective
    g_sdf = done["g_sdf"] if done else torch.empty(P, device=dev)
    g_rgb = done["g_rgb"] if done else torch.empty(P, 3, device=dev)
    g_grad = done["g_grad"] if done else torch.empty(P, 3, device=dev)
    if done:
        pass
    elif track is not None:
        with _timed("k_composite_track", P * 60):
            check(lib.nsa_composite_track(rays_o.data_ptr(), rays_d.data_ptr(), z_vals.data_ptr(), b["sdf"].data_ptr(),
                                          b["rgb"].data_ptr(), b["vox"].data_ptr(), model.voxel_res, R, S, track["gt"].data_ptr(),
                                          R, b["rgb_values"].data_ptr(), track["ray_loss"].data_ptr(), g_sdf.data_ptr(),
                                          g_rgb.data_ptr(), g_grad.data_ptr(), st))
    else:
        with _timed("k_composite_bwd", P * 60):
            check(lib.nsa_composite_backward(rays_o.data_ptr(), rays_d.data_ptr(), z_vals.data_ptr(), b["sdf"].data_ptr(),
                                             b["rgb"].data_ptr(), b["grad"].data_ptr(), b["vox"].data_ptr(), model.voxel_res,
                                             R, S, ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), ptr(gs[3]), ptr(gs[4]),
                                             g_sdf.data_ptr(), g_rgb.data_ptr(), g_grad.data_ptr(), st))
    g_feat = torch.empty(hl_size(P), device=dev)
    g_x = torch.empty(P, 3, device=dev)
    g_dir = torch.empty(P, 3, device=dev)
    grid_grad = 1 if color_stage != "base" else 0
    pg = {}
    want = params or {}
    # data-path backward (no parameter gradients) with the coarse network in the 32-point tiling: its backward rides in the colour
    # backward's launch (nsa_colour_coarse_backward)
    merged = (COLOUR_COARSE_BWD and not any(want.get(k) for k in ("flat_r", "tab_r", "flat_c", "tab_c")) and gc.tile != 16
              and gc.precision == gr.precision)
    if want.get("flat_r") or want.get("tab_r"):
        from . import mapping
        emit = mapping.new_emit(mapping.CE["ROWS"], P, dev) if want.get("flat_r") else None
        gt = _table_grad(model.rendering_network.encoding.embeddings) if want.get("tab_r") else None
        with _timed("k_colour_bwd<map>", P * 512):
            check(lib.nsa_colour_backward_params(ctypes.byref(pts), ctypes.byref(gr), pr.data_ptr(), b["grad"].data_ptr(),
                                                 b["feat"].data_ptr(), b["save"].data_ptr(), g_rgb.data_ptr(), grid_grad,
                                                 g_feat.data_ptr(), g_grad.data_ptr(), g_x.data_ptr(), g_dir.data_ptr(),
                                                 ptr(gt), ptr(emit), 0 if emit is None else emit.shape[1], st))
        if emit is not None:
            pg["flat_r"] = mapping.colour_flat_grad(emit)
            del emit
        pg["tab_r"] = _table_result(gt)
    elif merged:        # colour backward + coarse SDF backward: two phases of one launch (same 32-point tiles)
        with _timed("k_colour_coarse_bwd", P * (512 + 3 * 4 * 8 * 8 * 4)):
            check(lib.nsa_colour_coarse_backward(ctypes.byref(pts), ctypes.byref(gr), pr.data_ptr(), b["grad"].data_ptr(),
                                                 b["feat"].data_ptr(), b["save"].data_ptr(), g_rgb.data_ptr(), grid_grad,
                                                 g_feat.data_ptr(), g_grad.data_ptr(), g_x.data_ptr(), g_dir.data_ptr(),
                                                 ctypes.byref(gc), pc.data_ptr(), g_sdf.data_ptr(), st))
    else:
        with _timed("k_colour_bwd", P * 512):
            check(lib.nsa_colour_backward(ctypes.byref(pts), ctypes.byref(gr), pr.data_ptr(), b["grad"].data_ptr(),
                                          b["feat"].data_ptr(), b["save"].data_ptr(), g_rgb.data_ptr(), grid_grad,
                                          g_feat.data_ptr(), g_grad.data_ptr(), g_x.data_ptr(), g_dir.data_ptr(), st))
    if want.get("flat_c") or want.get("tab_c"):
        from . import mapping
        enc = imp.coarse.encoding
        # the MAP backward has its own tiling (the forward's results reach it in tiling-independent layouts only)
        tile_m = tile_of(model, "coarse_map")
        gcm, keep_cm = sdf_grid_desc(model, "coarse", "coarse_map")
        pcm = packed_sdf(model, "coarse", use="coarse_map")
        emit = mapping.new_emit(mapping.se_rows(1, tile_m)["ROWS"], P, dev, extra=2) if want.get("flat_c") else None
        gt = _table_grad(enc.embeddings) if want.get("tab_c") else None
        with _timed("k_sdfnet_bwd<coarse,map>", P * 3 * 4 * 8 * 8 * 4):
            check(lib.nsa_sdfnet_backward_params(ctypes.byref(pts), ctypes.byref(gcm), pcm.data_ptr(), g_sdf.data_ptr(),
                                                 g_feat.data_ptr(), g_grad.data_ptr(), 1, g_x.data_ptr(), ptr(gt),
                                                 ptr(emit), 0 if emit is None else emit.shape[1], st))
        if emit is not None:
            pg["flat_c"] = mapping.sdf_flat_grad(emit, g_sdf, P, enc.num_levels, enc.level_dim, tile=tile_m,
                                                 order=order)       # (emission columns are work items)
            del emit
        pg["tab_c"] = _table_result(gt)
    elif not merged:
        with _timed("k_sdfnet_bwd<coarse>", P * 3 * 4 * 8 * 8 * 4):
            check(lib.nsa_sdfnet_backward(ctypes.byref(pts), ctypes.byref(gc), pc.data_ptr(), g_sdf.data_ptr(),
                                          g_feat.data_ptr(), g_grad.data_ptr(), 1, g_x.data_ptr(), st))
    if stage != "coarse":
        if want.get("tab_f") or want.get("flat_f"):
            from . import mapping
            enc = imp.fine.encoding
            gt = _table_grad(enc.embeddings) if want.get("tab_f") else None
            emit = (mapping.new_emit(mapping.se_rows(3, tile_of(model, "fine"))["ROWS"], P, dev, extra=2)
                    if want.get("flat_f") else None)
            with _timed("k_sdfnet_bwd<fine,map>", P * 3 * 8 * 8 * 4 * 4):
                check(lib.nsa_sdfnet_backward_params(ctypes.byref(pts), ctypes.byref(gf), pf.data_ptr(), g_sdf.data_ptr(),
                                                     g_feat.data_ptr(), g_grad.data_ptr(), 1, g_x.data_ptr(),
                                                     ptr(gt), ptr(emit), 0 if emit is None else emit.shape[1], st))
            if emit is not None:
                pg["flat_f"] = mapping.sdf_flat_grad(emit, g_sdf, P, enc.num_levels, enc.level_dim, NH=3,
                                                     tile=tile_of(model, "fine"), order=order)
                del emit
            pg["tab_f"] = _table_result(gt)
        else:
            with _timed("k_sdfnet_bwd<fine>", P * 3 * 8 * 8 * 4 * 4):
                check(lib.nsa_sdfnet_backward(ctypes.byref(pts), ctypes.byref(gf), pf.data_ptr(), g_sdf.data_ptr(),
                                              g_feat.data_ptr(), g_grad.data_ptr(), 1, g_x.data_ptr(), st))
    if "_keep" in b:      # diagnostics (tools/diag_config0.py): per-point cotangents of the stages
        b["_keep"].update(g_x=g_x, g_dir=g_dir, g_sdf=g_sdf, g_rgb=g_rgb, g_grad=g_grad, g_feat=g_feat)
    if not reduce_rays:
        return g_x, g_dir
    g_o = torch.empty(R, 3, device=dev)
    g_d = torch.empty(R, 3, device=dev)
    check(lib.nsa_rays_backward(z_vals.data_ptr(), g_x.data_ptr(), g_dir.data_ptr(), R, S, g_o.data_ptr(),
                                g_d.data_ptr(), st))
    if params is not None:
        return g_o, g_d, pg
    return g_o, g_d


class FusedComposite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, z_vals, model, stage, color_stage):
        rays_o, rays_d, z_vals = rays_o.contiguous(), rays_d.contiguous(), z_vals.contiguous()
        R, S = z_vals.shape
        b = composite_forward_raw(model, rays_o, rays_d, z_vals, stage, any(ctx.needs_input_grad[:2]))
        ctx.save_for_backward(rays_o, rays_d, z_vals)
        ctx.bufs, ctx.model, ctx.stage, ctx.color_stage = b, model, stage, color_stage
        ctx.set_materialize_grads(False)      # outputs the loss does not use arrive as None (a NULL cotangent), not as zero fills
        sdf_o, rgb_o, grad_o = b["sdf"].view(R, S), b["rgb"].view(R, S, 3), b["grad"]
        ctx.mark_non_differentiable(sdf_o, rgb_o, grad_o)
        return b["rgb_values"], b["depth"].unsqueeze(-1), b["nmap"], b["weights"], b["entropy"], sdf_o, rgb_o, grad_o

    @staticmethod
    def backward(ctx, g_rgbv, g_depth, g_nmap, g_w, g_ent, *_unused):
        rays_o, rays_d, z_vals = ctx.saved_tensors
        g_o, g_d = composite_backward_raw(ctx.model, rays_o, rays_d, z_vals, ctx.bufs, ctx.stage, ctx.color_stage,
                                          g_rgbv, g_depth, g_nmap, g_ent, g_w)
        ctx.bufs = None
        return g_o, g_d, None, None, None, None


def composite(model, rays_o, rays_d, z_vals, stage, color_stage):
    return FusedComposite.apply(rays_o, rays_d, z_vals, model, stage, color_stage)


class FusedRays(torch.autograd.Function):
    """pose[b,4,4] (+ uv, K) -> rays_o[b*n,3], rays_d[b*n,3], depth_scale[b*n]; backward to the pose only."""

    @staticmethod
    def forward(ctx, pose, uv, K):
        pose, uv, K = pose.contiguous(), uv.contiguous(), K.contiguous()
        b, n, _ = uv.shape
        dev = uv.device
        rays_o = torch.empty(b * n, 3, device=dev)
        rays_d = torch.empty(b * n, 3, device=dev)
        depth_scale = torch.empty(b * n, device=dev)
        check(lib.nsa_rays_forward(uv.data_ptr(), pose.data_ptr(), K.data_ptr(), b, n, rays_o.data_ptr(),
                                   rays_d.data_ptr(), depth_scale.data_ptr(), _stream()))
        ctx.save_for_backward(pose, uv, K)
        ctx.mark_non_differentiable(depth_scale)
        return rays_o, rays_d, depth_scale

    @staticmethod
    def backward(ctx, g_o, g_d, _g_ds):
        pose, uv, K = ctx.saved_tensors
        b, n, _ = uv.shape
        zero = lambda g: torch.zeros(b * n, 3, device=uv.device) if g is None else g.contiguous()
        g_o, g_d = zero(g_o), zero(g_d)
        g_pose = torch.empty(b, 4, 4, device=uv.device)
        check(lib.nsa_rays_pose_backward(uv.data_ptr(), pose.data_ptr(), K.data_ptr(), b, n, g_o.data_ptr(),
                                         g_d.data_ptr(), g_pose.data_ptr(), _stream()))
        return g_pose, None, None


def rays(pose, uv, K):
    return FusedRays.apply(pose, uv, K)
