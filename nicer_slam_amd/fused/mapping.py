"""Fused mapping iteration: the composite pass and the eikonal-sample pass with PARAMETER gradients.

The backward kernels of fused/render.py (run as their MAP variants, Section 2 of the C ABI) accumulate the grid-table
gradients with run-merged atomics and write, per point, the vectors whose outer products are the MLP weight
gradients; the weight gradients themselves are a handful of [64 x n] x [n x ~130] products over those rows (nsa_emit_gemm,
csrc/emit_gemm.hip: K-chunked so that the long reduction dimension fills the chip, bias gradients as an extra column).

Trainable parameters follow the reference's optimizer list (code/training/volsdf_train.py:150-173): the three grid
tables, the coarse SDF MLP and the colour MLP.  The fine SDF MLP is pretrained and never handed to the optimizer there, yet
its parameters still require grad, so the reference computes gradients nobody reads; this engine produces them on request
(``model.fine_mlp_grads = True``: 976 emission rows from the fine MAP kernel) and skips that work by default.

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
from .render import composite_forward_raw, composite_backward_raw, hl_size, morton_order, _stream, _table_grad, _table_result
from .sampler import forward_pair_ok, grid_desc, packed_sdf, precision_of, sdf_grid_desc, tile_of

KCHUNK = 4096
SORT_POINTS = True      # run the per-point kernels of a mapping iteration in Morton order (see render.morton_order)


def se_rows(NH, tile=32):
    """Emission row map of an SDF network with NH hidden layers = struct SE<NH> of csrc/render_sdfnet.hip (tile 32) resp.
    SE4<NH> of csrc/render_sdfnet4.hip (tile 16):
    [H0 | TIN | DA_1.. | H_1.. | TH_1..TH_{NH-1} | AB_1..AB_NH | TH_NH | FB]; H0 / TIN hold one row per first-layer slot
    (72 resp. 96 rows, "IN"), the other regions 64 rows (hidden features in reference order)."""
    n_in = 96 if tile == 16 else 72
    b = 2 * n_in
    m = {"H0": 0, "TIN": n_in, "IN": n_in}
    for k in range(1, NH + 1):
        m[f"DA{k}"] = b + 64 * (k - 1)
        m[f"H{k}"] = b + 64 * NH + 64 * (k - 1)
        m[f"AB{k}"] = b + 128 * NH + 64 * (NH - 1) + 64 * (k - 1)
        if k < NH:
            m[f"TH{k}"] = b + 128 * NH + 64 * (k - 1)
    m[f"TH{NH}"] = m["AB1"] + 64 * NH
    m["FB"] = m[f"TH{NH}"] + 64
    m["ROWS"] = m["FB"] + 64
    return m


SE = se_rows(1)      # coarse network, 32-point tiling (464 rows); fine: se_rows(3) (976 rows); quad tiling: 512 / 1024
assert SE == dict(H0=0, TIN=72, IN=72, DA1=144, H1=208, AB1=272, TH1=336, FB=400, ROWS=464)
assert se_rows(1, 16)["ROWS"] == 512 and se_rows(3, 16)["ROWS"] == 1024
CE = dict(IN=0, H1=130, H2=194, AB1=258, AB2=322, OB=386, ROWS=389)              # = enum CE_* (render_colour.hip)


@functools.lru_cache(maxsize=None)
def _zero1(device):
    """a [1] zero on ``device``, made once (the flat gradient vectors end in one; it is only ever read)"""
    return torch.zeros(1, device=device)


def emit_ld(P):
    return ((P + KCHUNK - 1) // KCHUNK) * KCHUNK


def new_emit(rows, P, device, extra=0):
    """Emission buffer [rows + extra][ld]; the columns past the last (16-point) tile are never written by the kernels.  ``extra``
    rows past the kernels' own are written by nsa_emit_row (sdf_flat_grad: a row of ones and the per-point sdf cotangent)."""
    ld = emit_ld(P)
    buf = torch.empty(rows + extra, ld, device=device)
    tail = ((P + 15) // 16) * 16
    if tail < ld:
        buf[:rows, tail:].zero_()
    return buf


def emit_gemm(emit, a_rows, M, b_rows, N, sums=True, workspace=None):
    """out[M, N + sums] = sum over the (one or two) row pairs of emit[a : a + M] @ emit[b : b + N]^T, last column = row sums of
    the first A block: one weight gradient with its bias gradient (nsa_emit_gemm: fp32-faithful MFMA products straight from the
    emission rows, deterministic chunk-order reduction)."""
    rows, ld = emit.shape
    pairs = len(a_rows)
    b_rows = tuple(b_rows) if N else (0,) * pairs
    assert pairs in (1, 2) and len(b_rows) == pairs and emit.is_contiguous() and emit.dtype == torch.float32
    assert all(a + M <= rows for a in a_rows) and all(b + N <= rows for b in b_rows)
    need = int(lib.nsa_emit_gemm_workspace(ld, M, N, int(sums)))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=emit.device)
    out = torch.empty(M, N + int(sums), device=emit.device)
    check(lib.nsa_emit_gemm(emit.data_ptr(), ld, pairs, (ctypes.c_uint32 * 2)(*a_rows, *([0] * (2 - pairs))),
                            (ctypes.c_uint32 * 2)(*b_rows, *([0] * (2 - pairs))), M, N, int(sums), out.data_ptr(),
                            workspace.data_ptr(), _stream()))
    return out


def _workspace(emit, M=64, N=159):
    return torch.empty(int(lib.nsa_emit_gemm_workspace(emit.shape[1], M, N, 1)), device=emit.device)


@functools.lru_cache(maxsize=None)
def _sdf_rows(L, C, tile=32):
    """row of the H0 / TIN regions holding reference input feature f = 0..70: 2*slot + half (32-point tiling) resp.
    4*slot + quarter (quad tiling)."""
    rows = np.full(39 + L * C, -1, dtype=np.int64)
    if tile == 16:
        for s in range(pack.QIN_STEPS):
            for q in range(4):
                f = pack.sdf_in_feature4(s, q, C)
                if f >= 0:
                    rows[f] = 4 * s + q
    else:
        for s in range(pack.SDF_IN_STEPS):
            for h in range(2):
                f = pack.sdf_in_feature(s, h, L, C)
                if f >= 0:
                    rows[f] = 2 * s + h
    assert (rows >= 0).all()
    return torch.from_numpy(rows)


@functools.lru_cache(maxsize=None)
def _on(device, fn, *args):
    """device-resident copy of a (cached) host index map: uploaded once, not per iteration (a pageable upload is synchronous)"""
    return fn(*args).to(device)


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


def sdf_flat_grad(emit, g_sdf, P, L, C, NH=1, tile=32, order=None):
    """Gradient of an SDF network's flat parameter vector [W0(64x71), b0, W1, b1, .., W_NH(65x64), b_NH, 0] from its emission
    rows (row map and formulas: struct SE<NH>, csrc/render_sdfnet.hip; SE4<NH>, csrc/render_sdfnet4.hip).  ``g_sdf`` [P]: the
    per-point cotangent of the sdf value in POINT order, ``order`` the launch order of the emission columns (None: identity);
    ``emit`` then carries two rows past the kernels' own (new_emit(..., extra=2))."""
    m = se_rows(NH, tile)
    IN, ws = m["IN"], _workspace(emit)
    parts = []
    # first layer: value path (AB_1 x H0) + its share of the reverse pass (DA_1 x TIN); the last column is the bias gradient
    W0 = emit_gemm(emit, (m["AB1"], m["DA1"]), 64, (m["H0"], m["TIN"]), IN, workspace=ws)
    parts += [W0[:, :IN][:, _on(emit.device, _sdf_rows, L, C, tile)].reshape(-1), W0[:, IN]]
    for k in range(1, NH):                                   # hidden layer k, same two paths
        Wk = emit_gemm(emit, (m[f"AB{k + 1}"], m[f"DA{k + 1}"]), 64, (m[f"H{k}"], m[f"TH{k}"]), 64, workspace=ws)
        parts += [Wk[:, :64].reshape(-1), Wk[:, 64]]
    if g_sdf is None:
        row0 = emit_gemm(emit, (m[f"TH{NH}"],), 64, None, 0, workspace=ws)[:, 0]       # sdf row: row sums of TH_NH
        dbs = _zero1(emit.device)
    else:
        # sdf row = row sums of TH_NH + H_NH g_sdf: one product pair against two rows written here (ones; g_sdf in launch order)
        ROWS, ld = m["ROWS"], emit.shape[1]
        assert emit.shape[0] >= ROWS + 2 and g_sdf.is_contiguous() and g_sdf.numel() == P
        st = _stream()
        check(lib.nsa_emit_row(emit[ROWS].data_ptr(), None, None, P, ld, 1.0, st))
        check(lib.nsa_emit_row(emit[ROWS + 1].data_ptr(), g_sdf.data_ptr(), None if order is None else order.data_ptr(), P, ld,
                               0.0, st))
        row0 = emit_gemm(emit, (m[f"TH{NH}"], m[f"H{NH}"]), 64, (ROWS, ROWS + 1), 1, sums=False, workspace=ws)[:, 0]
        dbs = g_sdf.sum().reshape(1)
    Wf = emit_gemm(emit, (m["FB"],), 64, (m[f"H{NH}"],), 64, workspace=ws)               # feature rows + their biases
    parts += [row0, Wf[:, :64].reshape(-1), dbs, Wf[:, 64], _zero1(emit.device)]
    return torch.cat(parts)


def colour_flat_grad(emit):
    """Gradient of the colour network's flat parameter vector [W0(64x129), b0, W1, b1, W2(3x64), b2, 0]."""
    ws = _workspace(emit)
    W0 = emit_gemm(emit, (CE["AB1"],), 64, (CE["IN"],), 130, workspace=ws)
    W1 = emit_gemm(emit, (CE["AB2"],), 64, (CE["H1"],), 64, workspace=ws)
    W2 = emit_gemm(emit, (CE["OB"],), 3, (CE["H2"],), 64, workspace=ws)
    return torch.cat([W0[:, :130][:, _on(emit.device, _col_rows)].reshape(-1), W0[:, 130], W1[:, :64].reshape(-1), W1[:, 64],
                      W2[:, :64].reshape(-1), W2[:, 64], _zero1(emit.device)])


def _nets(model):
    imp = model.implicit_network
    return imp.coarse, imp.fine, model.rendering_network


def fine_mlp_wanted(model):
    """True when the fine SDF MLP's gradients are to be produced: its parameters require grad AND
    ``model.fine_mlp_grads`` is set.  The reference computes them and never applies them (the fine MLP is pretrained and not
    in the optimizer, volsdf_train.py:140-173), so the default skips that work; see SLAMNetwork.fine_mlp_grads."""
    return bool(getattr(model, "fine_mlp_grads", False)) and any(p.requires_grad for p in _nets(model)[1].mlp_parameters())


def params_supported(model):
    """True when every parameter that requires grad is one this engine produces a gradient for -- or is the fine SDF MLP,
    whose gradients are produced on request only (fine_mlp_wanted)."""
    c, f, r = _nets(model)
    covered = {id(p) for p in list(c.mlp_parameters()) + list(c.grid_parameters()) + list(f.grid_parameters())
               + list(f.mlp_parameters()) + list(r.mlp_parameters()) + list(r.grid_parameters())}
    return all((not p.requires_grad) or id(p) in covered for p in model.parameters())


class FusedCompositeParams(torch.autograd.Function):
    """FusedComposite with gradients for (flat coarse MLP, flat colour MLP, coarse / fine / colour tables)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, z_vals, flat_c, flat_r, tab_c, tab_f, tab_r, flat_f, model, stage, color_stage):
        rays_o, rays_d, z_vals = rays_o.contiguous(), rays_d.contiguous(), z_vals.contiguous()
        R, S = z_vals.shape
        b = composite_forward_raw(model, rays_o, rays_d, z_vals, stage, True, sort_points=SORT_POINTS)
        ctx.save_for_backward(rays_o, rays_d, z_vals)
        ctx.bufs, ctx.model, ctx.stage, ctx.color_stage = b, model, stage, color_stage
        sdf_o, rgb_o, grad_o = b["sdf"].view(R, S), b["rgb"].view(R, S, 3), b["grad"]
        ctx.mark_non_differentiable(sdf_o, rgb_o, grad_o)
        ctx.set_materialize_grads(False)      # outputs the loss does not use arrive as None (a NULL cotangent), not as zero fills
        return b["rgb_values"], b["depth"].unsqueeze(-1), b["nmap"], b["weights"], b["entropy"], sdf_o, rgb_o, grad_o

    @staticmethod
    def backward(ctx, g_rgbv, g_depth, g_nmap, g_w, g_ent, *_unused):
        rays_o, rays_d, z_vals = ctx.saved_tensors
        need = ctx.needs_input_grad
        want = dict(flat_c=need[3], flat_r=need[4], tab_c=need[5], tab_f=need[6] and ctx.stage != "coarse",
                    tab_r=need[7] and ctx.color_stage != "base", flat_f=need[8] and ctx.stage != "coarse")
        g_o, g_d, pg = composite_backward_raw(ctx.model, rays_o, rays_d, z_vals, ctx.bufs, ctx.stage, ctx.color_stage,
                                              g_rgbv, g_depth, g_nmap, g_ent, g_w, params=want)
        ctx.bufs = None
        return (g_o, g_d, None, pg.get("flat_c"), pg.get("flat_r"), pg.get("tab_c"), pg.get("tab_f"), pg.get("tab_r"),
                pg.get("flat_f"), None, None, None)


class FusedSdfGradient(torch.autograd.Function):
    """grad sdf at explicit points (ImplicitNetworkGrid_COMBINE.gradient, base_networks.py:37-47 -- the eikonal samples
    of network.py:313-336) with gradients for (flat coarse MLP, coarse table, fine table)."""

    @staticmethod
    def forward(ctx, points, flat_c, tab_c, tab_f, flat_f, model, stage, halves=False):
        points = points.contiguous()
        N = points.shape[0]
        halves = bool(halves) and N % 2 == 0
        ctx.halves = halves
        dev = points.device
        imp = model.implicit_network
        gc, keep_c = sdf_grid_desc(model, "coarse")
        gf, keep_f = sdf_grid_desc(model, "fine")
        pc, pf = packed_sdf(model, "coarse"), packed_sdf(model, "fine")
        order = morton_order(PointsDesc(None, None, None, points.data_ptr(), N, 0, None), N, dev) if SORT_POINTS else None
        pts = PointsDesc(None, None, None, points.data_ptr(), N, 0, None if order is None else order.data_ptr())
        sdf = torch.empty(N, device=dev)
        grad = torch.empty(N, 3, device=dev)
        feat = torch.empty(hl_size(N), device=dev)
        st = _stream()
        if stage != "coarse" and forward_pair_ok(model):
            gcp, keep_cp = sdf_grid_desc(model, "coarse", "coarse_pair")
            pcp = packed_sdf(model, "coarse", use="coarse_pair")
            with _timed("k_sdfnet_fwd<pair,eik>", 0):
                check(lib.nsa_sdfnet_forward_pair(ctypes.byref(pts), ctypes.byref(gcp), ctypes.byref(gf), pcp.data_ptr(),
                                                  pf.data_ptr(), sdf.data_ptr(), grad.data_ptr(), feat.data_ptr(), st))
        else:
            with _timed("k_sdfnet_fwd<coarse,eik>", 0):
                check(lib.nsa_sdfnet_forward(ctypes.byref(pts), ctypes.byref(gc), pc.data_ptr(), 0, sdf.data_ptr(),
                                             grad.data_ptr(), feat.data_ptr(), st))
            if stage != "coarse":
                with _timed("k_sdfnet_fwd<fine,eik>", 0):
                    check(lib.nsa_sdfnet_forward(ctypes.byref(pts), ctypes.byref(gf), pf.data_ptr(), 1, sdf.data_ptr(),
                                                 grad.data_ptr(), feat.data_ptr(), st))
        ctx.save_for_backward(points)
        ctx.model, ctx.stage, ctx.packs, ctx.order = model, stage, (pc, pf), order
        if halves:          # the two halves as two outputs: the loss's cotangents come back as two tensors, not through slice_backward
            ctx.set_materialize_grads(False)
            return grad[:N // 2], grad[N // 2:]
        return grad

    @staticmethod
    def backward(ctx, g, g2=None):
        (points,) = ctx.saved_tensors
        if ctx.halves:
            n_half = points.shape[0] // 2
            if g is None and g2 is None:
                return (None,) * 8
            if (g is not None and g2 is not None and g.is_contiguous() and g2.is_contiguous()
                    and g2.data_ptr() == g.data_ptr() + g.numel() * g.element_size()
                    and g.untyped_storage().data_ptr() == g2.untyped_storage().data_ptr()):
                g = torch.as_strided(g, (2 * n_half, 3), (3, 1))          # adjacent halves of one buffer (fused/loss.py): no copy
            else:
                z = lambda t: torch.zeros(n_half, 3, device=points.device) if t is None else t
                g = torch.cat([z(g), z(g2)], 0)
        model, stage = ctx.model, ctx.stage
        N = points.shape[0]
        dev = points.device
        imp = model.implicit_network
        gc, keep_c = sdf_grid_desc(model, "coarse")
        gf, keep_f = sdf_grid_desc(model, "fine")
        pc, pf = ctx.packs
        order = ctx.order
        pts = PointsDesc(None, None, None, points.data_ptr(), N, 0, None if order is None else order.data_ptr())
        g = g.contiguous()
        g_x = torch.empty(N, 3, device=dev)
        need = ctx.needs_input_grad
        st = _stream()
        out = [None, None, None, None, None, None, None, None]
        tile = tile_of(model, "coarse_map")
        emit = new_emit(se_rows(1, tile)["ROWS"], N, dev) if need[1] else None
        gt_c = _table_grad(imp.coarse.encoding.embeddings) if need[2] else None
        if emit is not None or gt_c is not None:
            gcm, keep_cm = sdf_grid_desc(model, "coarse", "coarse_map")
            pcm = packed_sdf(model, "coarse", use="coarse_map")
            with _timed("k_sdfnet_bwd<coarse,eik>", 0):
                check(lib.nsa_sdfnet_backward_params(ctypes.byref(pts), ctypes.byref(gcm), pcm.data_ptr(), None, None,
                                                     g.data_ptr(), 0, g_x.data_ptr(),
                                                     None if gt_c is None else gt_c.data_ptr(),
                                                     None if emit is None else emit.data_ptr(),
                                                     0 if emit is None else emit.shape[1], st))
            if emit is not None:
                enc = imp.coarse.encoding
                out[1] = sdf_flat_grad(emit, None, N, enc.num_levels, enc.level_dim, tile=tile)
            out[2] = _table_result(gt_c)
        if (need[3] or need[4]) and stage != "coarse":
            gt_f = _table_grad(imp.fine.encoding.embeddings) if need[3] else None
            emit_f = new_emit(se_rows(3, tile_of(model, "fine"))["ROWS"], N, dev) if need[4] else None
            with _timed("k_sdfnet_bwd<fine,eik>", 0):
                check(lib.nsa_sdfnet_backward_params(ctypes.byref(pts), ctypes.byref(gf), pf.data_ptr(), None, None,
                                                     g.data_ptr(), 0, g_x.data_ptr(),
                                                     None if gt_f is None else gt_f.data_ptr(),
                                                     None if emit_f is None else emit_f.data_ptr(),
                                                     0 if emit_f is None else emit_f.shape[1], st))
            out[3] = _table_result(gt_f)
            if emit_f is not None:
                enc = imp.fine.encoding
                out[4] = sdf_flat_grad(emit_f, None, N, enc.num_levels, enc.level_dim, NH=3, tile=tile_of(model, "fine"))
        return tuple(out)


def flat_inputs(model):
    """(flat coarse MLP, flat colour MLP, coarse table, fine table, colour table, flat fine MLP or None) as autograd
    inputs; the last one only when its gradients are wanted (fine_mlp_wanted).  Inside one SLAMNetwork.forward the composite pass
    and the eikonal pass share them (``model._flat_inputs_fwd``, set and cleared by forward): one weight-norm node per network."""
    shared = model.__dict__.get("_flat_inputs_fwd")
    if shared is not None and shared[0] == torch.is_grad_enabled():
        return shared[1]
    c, f, r = _nets(model)
    out = (pack.flat_params(c), pack.flat_params(r), c.encoding.embeddings, f.encoding.embeddings,
           r.encoding.embeddings, pack.flat_params(f) if fine_mlp_wanted(model) else None)
    if "_flat_inputs_fwd" in model.__dict__:
        model.__dict__["_flat_inputs_fwd"] = (torch.is_grad_enabled(), out)
    return out


def composite(model, rays_o, rays_d, z_vals, stage, color_stage):
    flat_c, flat_r, tab_c, tab_f, tab_r, flat_f = flat_inputs(model)
    return FusedCompositeParams.apply(rays_o, rays_d, z_vals, flat_c, flat_r, tab_c, tab_f, tab_r, flat_f, model, stage,
                                      color_stage)


def sdf_gradient(model, points, stage, halves=False):
    """grad sdf at ``points``; ``halves``: returned as (first half, second half) -- the eikonal samples and their jittered neighbours
    (network.py:313-336) -- as two outputs of the Function."""
    flat_c, _, tab_c, tab_f, _, flat_f = flat_inputs(model)
    return FusedSdfGradient.apply(points, flat_c, tab_c, tab_f, flat_f, model, stage, halves)


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
