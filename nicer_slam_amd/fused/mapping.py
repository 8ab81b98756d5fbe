"""Fused mapping iteration: the composite pass and the eikonal-sample pass with PARAMETER gradients.

The backward kernels of fused/render.py (run as their MAP variants, Section 2 of the C ABI) accumulate the grid-table
gradients with run-merged atomics and write, per point, the vectors whose outer products are the MLP weight
gradients; the weight gradients themselves are a handful of [64 x n] x [n x ~130] GEMMs over those rows, batched here
in K-chunks so that the long reduction dimension fills the chip.

Trainable parameters follow the reference's optimizer list (code/training/volsdf_train.py:150-173): the three grid
tables, the coarse SDF MLP and the colour MLP.  The fine SDF MLP is pretrained and frozen there; this engine does not
produce gradients for it (SLAMNetwork.freeze_fine_mlp()).

Autograd contract: both Functions take the FLAT effective parameter vectors of pack.flat_params() (weight-norm already
applied, differentiably) and return gradients in the same layout, so torch carries them on to weight_g / weight_v /
bias.  Replaces torch.autograd through base_networks.py:195-221, 333-395 and hashgrid.py:64-141 for one mapping
iteration (reference code/model/network.py:112-151, 313-345).
"""
import ctypes
import functools

import numpy as np
import torch

from .._native import lib, check, PointsDesc
from ..hashencoder.backend import _timed
from . import pack
from .render import composite_forward_raw, composite_backward_raw, hl_size, morton_order, _stream
from .sampler import grid_desc, packed_sdf, precision_of

KCHUNK = 4096
SORT_POINTS = True      # run the per-point kernels of a mapping iteration in Morton order (see render.morton_order)
SE = dict(H0=0, TIN=72, DA1=144, H1=208, AB1=272, TH1=336, FB=400, ROWS=464)      # = enum SE_* (render_sdfnet.hip)
CE = dict(IN=0, H1=130, H2=194, AB1=258, AB2=322, OB=386, ROWS=389)              # = enum CE_* (render_colour.hip)


def emit_ld(P):
    return ((P + KCHUNK - 1) // KCHUNK) * KCHUNK


def new_emit(rows, P, device):
    """Emission buffer [rows][ld]; the columns past the last 32-point tile are never written by the kernel."""
    ld = emit_ld(P)
    buf = torch.empty(rows, ld, device=device)
    tail = ((P + 31) // 32) * 32
    if tail < ld:
        buf[:, tail:].zero_()
    return buf


def outer_sum(A, B):
    """A[m, ld] @ B[n, ld]^T with the reduction split into ld/KCHUNK batches (strided views, no copies)."""
    m, ld = A.shape
    n = B.shape[0]
    nch = ld // KCHUNK
    A3 = A.view(m, nch, KCHUNK).transpose(0, 1)
    B3 = B.view(n, nch, KCHUNK).transpose(0, 1)
    return torch.bmm(A3, B3.transpose(1, 2)).sum(0)


@functools.lru_cache(maxsize=None)
def _sdf_rows(L, C):
    """row (2*slot + half) of the H0 / TIN regions holding reference input feature f = 0..70."""
    rows = np.full(39 + L * C, -1, dtype=np.int64)
    for s in range(pack.SDF_IN_STEPS):
        for h in range(2):
            f = pack.sdf_in_feature(s, h, L, C)
            if f >= 0:
                rows[f] = 2 * s + h
    assert (rows >= 0).all()
    return torch.from_numpy(rows)


@functools.lru_cache(maxsize=None)
def _col_rows():
    rows = np.full(129, -1, dtype=np.int64)
    for s in range(pack.COL_IN_STEPS):
        for h in range(2):
            f = pack.col_in_feature(s, h)
            if f >= 0:
                rows[f] = 2 * s + h
    assert (rows >= 0).all()
    return torch.from_numpy(rows)


def sdf_flat_grad(emit, g_sdf, P, L, C):
    """Gradient of the coarse network's flat parameter vector [W0(64x71), b0, W1(65x64), b1, 0] from its emission rows."""
    r = lambda name, n: emit[SE[name]:SE[name] + n]
    H0, TIN, AB1, DA1, H1, TH1, FB = r("H0", 72), r("TIN", 72), r("AB1", 64), r("DA1", 64), r("H1", 64), r("TH1", 64), r("FB", 64)
    M = outer_sum(AB1, H0) + outer_sum(DA1, TIN)
    dW0 = M[:, _sdf_rows(L, C).to(emit.device)]
    sums = emit[SE["AB1"]:SE["ROWS"]].sum(1)                 # AB1 | TH1 | FB row sums in one reduction
    db0, row0, fb_sum = sums[:64], sums[64:128], sums[128:]
    dbs = emit.new_zeros(1)
    if g_sdf is not None:
        row0 = row0 + H1[:, :P] @ g_sdf
        dbs = g_sdf.sum().reshape(1)
    dW1 = torch.cat([row0.unsqueeze(0), outer_sum(FB, H1)], 0)
    db1 = torch.cat([dbs, fb_sum])
    return torch.cat([dW0.reshape(-1), db0, dW1.reshape(-1), db1, emit.new_zeros(1)])


def colour_flat_grad(emit):
    """Gradient of the colour network's flat parameter vector [W0(64x129), b0, W1, b1, W2(3x64), b2, 0]."""
    r = lambda name, n: emit[CE[name]:CE[name] + n]
    IN, AB1, H1, AB2, H2, OB = r("IN", 130), r("AB1", 64), r("H1", 64), r("AB2", 64), r("H2", 64), r("OB", 3)
    dW0 = outer_sum(AB1, IN)[:, _col_rows().to(emit.device)]
    dW1 = outer_sum(AB2, H1)
    dW2 = outer_sum(OB, H2)
    sums = emit[CE["AB1"]:CE["ROWS"]].sum(1)                 # AB1 | AB2 | OB row sums in one reduction
    return torch.cat([dW0.reshape(-1), sums[:64], dW1.reshape(-1), sums[64:128], dW2.reshape(-1), sums[128:],
                      emit.new_zeros(1)])


def _nets(model):
    imp = model.implicit_network
    return imp.coarse, imp.fine, model.rendering_network


def params_supported(model):
    """True when every parameter that requires grad is one this engine produces a gradient for."""
    c, f, r = _nets(model)
    covered = {id(p) for p in list(c.mlp_parameters()) + list(c.grid_parameters()) + list(f.grid_parameters())
               + list(r.mlp_parameters()) + list(r.grid_parameters())}
    return all((not p.requires_grad) or id(p) in covered for p in model.parameters())


class FusedCompositeParams(torch.autograd.Function):
    """FusedComposite with gradients for (flat coarse MLP, flat colour MLP, coarse / fine / colour tables)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, z_vals, flat_c, flat_r, tab_c, tab_f, tab_r, model, stage, color_stage):
        rays_o, rays_d, z_vals = rays_o.contiguous(), rays_d.contiguous(), z_vals.contiguous()
        R, S = z_vals.shape
        b = composite_forward_raw(model, rays_o, rays_d, z_vals, stage, True, sort_points=SORT_POINTS)
        ctx.save_for_backward(rays_o, rays_d, z_vals)
        ctx.bufs, ctx.model, ctx.stage, ctx.color_stage = b, model, stage, color_stage
        sdf_o, rgb_o, grad_o = b["sdf"].view(R, S), b["rgb"].view(R, S, 3), b["grad"]
        ctx.mark_non_differentiable(sdf_o, rgb_o, grad_o)
        return b["rgb_values"], b["depth"].unsqueeze(-1), b["nmap"], b["weights"], b["entropy"], sdf_o, rgb_o, grad_o

    @staticmethod
    def backward(ctx, g_rgbv, g_depth, g_nmap, g_w, g_ent, *_unused):
        rays_o, rays_d, z_vals = ctx.saved_tensors
        need = ctx.needs_input_grad
        want = dict(flat_c=need[3], flat_r=need[4], tab_c=need[5], tab_f=need[6] and ctx.stage != "coarse",
                    tab_r=need[7] and ctx.color_stage != "base")
        g_o, g_d, pg = composite_backward_raw(ctx.model, rays_o, rays_d, z_vals, ctx.bufs, ctx.stage, ctx.color_stage,
                                              g_rgbv, g_depth, g_nmap, g_ent, g_w, params=want)
        ctx.bufs = None
        return (g_o, g_d, None, pg.get("flat_c"), pg.get("flat_r"), pg.get("tab_c"), pg.get("tab_f"), pg.get("tab_r"),
                None, None, None)


class FusedSdfGradient(torch.autograd.Function):
    """grad sdf at explicit points (ImplicitNetworkGrid_COMBINE.gradient, base_networks.py:37-47 -- the eikonal samples
    of network.py:313-336) with gradients for (flat coarse MLP, coarse table, fine table)."""

    @staticmethod
    def forward(ctx, points, flat_c, tab_c, tab_f, model, stage):
        points = points.contiguous()
        N = points.shape[0]
        dev = points.device
        imp = model.implicit_network
        gc, keep_c = grid_desc(imp.coarse.encoding, imp.coarse.divide_factor, 1, precision_of(model, "sdf"))
        gf, keep_f = grid_desc(imp.fine.encoding, imp.fine.divide_factor, 3, precision_of(model, "sdf"))
        pc, pf = packed_sdf(model, "coarse"), packed_sdf(model, "fine")
        order = morton_order(PointsDesc(None, None, None, points.data_ptr(), N, 0, None), N, dev) if SORT_POINTS else None
        pts = PointsDesc(None, None, None, points.data_ptr(), N, 0, None if order is None else order.data_ptr())
        sdf = torch.empty(N, device=dev)
        grad = torch.empty(N, 3, device=dev)
        feat = torch.empty(hl_size(N), device=dev)
        st = _stream()
        with _timed("k_sdfnet_fwd<coarse,eik>", 0):
            check(lib.nsa_sdfnet_forward(ctypes.byref(pts), ctypes.byref(gc), pc.data_ptr(), 0, sdf.data_ptr(),
                                         grad.data_ptr(), feat.data_ptr(), st))
        if stage != "coarse":
            with _timed("k_sdfnet_fwd<fine,eik>", 0):
                check(lib.nsa_sdfnet_forward(ctypes.byref(pts), ctypes.byref(gf), pf.data_ptr(), 1, sdf.data_ptr(),
                                             grad.data_ptr(), feat.data_ptr(), st))
        ctx.save_for_backward(points)
        ctx.model, ctx.stage, ctx.packs, ctx.order = model, stage, (pc, pf), order
        return grad

    @staticmethod
    def backward(ctx, g):
        (points,) = ctx.saved_tensors
        model, stage = ctx.model, ctx.stage
        N = points.shape[0]
        dev = points.device
        imp = model.implicit_network
        gc, keep_c = grid_desc(imp.coarse.encoding, imp.coarse.divide_factor, 1, precision_of(model, "sdf"))
        gf, keep_f = grid_desc(imp.fine.encoding, imp.fine.divide_factor, 3, precision_of(model, "sdf"))
        pc, pf = ctx.packs
        order = ctx.order
        pts = PointsDesc(None, None, None, points.data_ptr(), N, 0, None if order is None else order.data_ptr())
        g = g.contiguous()
        g_x = torch.empty(N, 3, device=dev)
        need = ctx.needs_input_grad
        st = _stream()
        out = [None, None, None, None, None, None]
        emit = new_emit(SE["ROWS"], N, dev) if need[1] else None
        gt_c = torch.zeros_like(imp.coarse.encoding.embeddings) if need[2] else None
        if emit is not None or gt_c is not None:
            with _timed("k_sdfnet_bwd<coarse,eik>", 0):
                check(lib.nsa_sdfnet_backward_params(ctypes.byref(pts), ctypes.byref(gc), pc.data_ptr(), None, None,
                                                     g.data_ptr(), 0, g_x.data_ptr(),
                                                     None if gt_c is None else gt_c.data_ptr(),
                                                     None if emit is None else emit.data_ptr(),
                                                     0 if emit is None else emit.shape[1], st))
            if emit is not None:
                enc = imp.coarse.encoding
                out[1] = sdf_flat_grad(emit, None, N, enc.num_levels, enc.level_dim)
            out[2] = gt_c
        if need[3] and stage != "coarse":
            gt_f = torch.zeros_like(imp.fine.encoding.embeddings)
            with _timed("k_sdfnet_bwd<fine,eik>", 0):
                check(lib.nsa_sdfnet_backward_params(ctypes.byref(pts), ctypes.byref(gf), pf.data_ptr(), None, None,
                                                     g.data_ptr(), 0, g_x.data_ptr(), gt_f.data_ptr(), None, 0, st))
            out[3] = gt_f
        return tuple(out)


def flat_inputs(model):
    """(flat coarse MLP, flat colour MLP, coarse table, fine table, colour table) as autograd inputs."""
    c, f, r = _nets(model)
    return (pack.flat_params(c), pack.flat_params(r), c.encoding.embeddings, f.encoding.embeddings,
            r.encoding.embeddings)


def composite(model, rays_o, rays_d, z_vals, stage, color_stage):
    flat_c, flat_r, tab_c, tab_f, tab_r = flat_inputs(model)
    return FusedCompositeParams.apply(rays_o, rays_d, z_vals, flat_c, flat_r, tab_c, tab_f, tab_r, model, stage,
                                      color_stage)


def sdf_gradient(model, points, stage):
    flat_c, _, tab_c, tab_f, _ = flat_inputs(model)
    return FusedSdfGradient.apply(points, flat_c, tab_c, tab_f, model, stage)


def update_voxels(model, rays_o, rays_d, z_vals):
    """Visit counter of the batch's samples, in place (SLAMNetwork.update_voxels, network.py:62-76)."""
    rays_o, rays_d, z_vals = rays_o.contiguous(), rays_d.contiguous(), z_vals.contiguous()
    R, S = z_vals.shape
    vox = model.voxels
    if not (vox.is_contiguous() and vox.dtype == torch.float32):
        raise RuntimeError("voxel counter must be a contiguous float32 tensor")
    pts = PointsDesc(rays_o.data_ptr(), rays_d.data_ptr(), z_vals.data_ptr(), None, R * S, S)
    with _timed("k_update_voxels", R * S * 16):
        check(lib.nsa_update_voxels(ctypes.byref(pts), vox.data_ptr(), model.voxel_res, _stream()))
