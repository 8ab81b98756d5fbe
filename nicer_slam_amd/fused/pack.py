"""Pack MLP parameters into MFMA fragment order for the fused kernels (layout: csrc/mlp_common.hpp, csrc/sdf_net.hpp).

Every packed block is a gather of the flattened effective parameters (weight-norm applied: W = g * v / |v|_row,
reference base_networks.py:148-149); weight blocks are additionally split exactly into three bfloat16 pieces
(hi + mid + lo = the fp32 weight) for the split-bf16 GEMM.  Index maps are built once per network shape with numpy;
packing runs on the device and is cached on the parameters' version counters.
"""
import functools

import numpy as np
import torch

SDF_IN_STEPS = 36
COL_IN_STEPS = 65


def F(r, h):
    """MFMA 32x32 result row of register r in half-wave h."""
    return (r & 3) + 8 * (r >> 2) + 4 * h


def hid_feature(s, h):
    """hidden feature held by k-step s of half-wave h."""
    return 32 * (s >> 4) + F(s & 15, h)


def sdf_in_feature(s, h, L, C):
    """reference input-feature index (base_networks.py:155-164: [x, PE6(x), grid]) of first-layer slot (s, h)."""
    if s == 0:
        return 0 if h == 0 else 2
    if s == 1:
        return 1 if h == 0 else -1
    if s < 20:
        j, which = divmod(s - 2, 2)
        g = 2 * j + h
        k, d = divmod(g, 3)
        return 3 + 6 * k + d + (3 if which else 0)
    jl, c = divmod(s - 20, C)
    return 39 + (2 * jl + h) * C + c


def row_slot(mt, i):
    """(slot, half) owned by output row i of tile mt in a GEMM whose OUTPUT is a slot list (transposed first layer)."""
    hh = (i >> 2) & 1
    r = (i & 3) + 4 * (i >> 3)
    return 16 * mt + r, hh


def a_block(MT, KS, elem):
    """int64 index block [MT][KS8][64 lanes][8]; elem(mt, i, s, h) -> flat parameter index or -1 (zero).
    (One index per fp32 weight; pack_blocks() splits the gathered values into the three bf16 pieces.)"""
    KS8 = (KS + 7) // 8
    out = np.full((MT, KS8, 64, 8), -1, dtype=np.int64)
    for mt in range(MT):
        for g in range(KS8):
            for lane in range(64):
                for e in range(8):
                    s = 8 * g + e
                    if s < KS:
                        out[mt, g, lane, e] = elem(mt, lane & 31, s, lane >> 5)
    return ("A", out.reshape(-1))


def vec_block(n_tiles, elem, pad_to=None):
    """activation-layout vector [(t*2+h)*16 + r]; elem(feature) -> flat index."""
    out = np.full(n_tiles * 32, -1, dtype=np.int64)
    for t in range(n_tiles):
        for h in range(2):
            for r in range(16):
                out[(t * 2 + h) * 16 + r] = elem(32 * t + F(r, h))
    if pad_to:
        out = np.concatenate([out, np.full(pad_to - out.size, -1, dtype=np.int64)])
    return ("V", out)


def split_bf16x3(x):
    """exact 3-way split x = hi + mid + lo into bfloat16 pieces (8+8+8 significand bits)."""
    hi = x.to(torch.bfloat16)
    r = x - hi.float()
    mid = r.to(torch.bfloat16)
    lo = (r - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


W_SCALE = 512.0      # csrc/mlp_common.hpp::kWScale


def split_f16x2(x):
    """form 2 of the packed weights (csrc/mlp_common.hpp::NSA_FORM): 512 x = h0 + h1 in two round-to-nearest fp16 pieces, plus the
    bf16 round-to-nearest value of x (third slot: what the bf16-operand kernels multiply with); 16-bit words as int16 views."""
    t = x * W_SCALE
    h0 = t.to(torch.float16)
    h1 = (t - h0.float()).to(torch.float16)
    return h0.view(torch.int16), h1.view(torch.int16), x.to(torch.bfloat16).view(torch.int16)


def operand_form():
    """3 (three bf16 pieces) or 2 (two fp16 pieces): what the loaded library's fp32 kernels multiply with (nsa_operand_form)."""
    from .._native import lib
    return int(lib.nsa_operand_form())


def split_pieces(x):
    """the three 16-bit pieces of every weight as the loaded library's kernels read them (int16 views; nsa_operand_form)"""
    if operand_form() == 2:
        return split_f16x2(x)
    return tuple(p.view(torch.int16) for p in split_bf16x3(x))


_plans = {}


def _plan(blocks, device):
    """Device-resident gather plan of one packed block list (built once per (list, device): the index maps never change, and
    uploading them per call cost 18 synchronous host->device copies per mapping iteration): the indices of all 'A' blocks
    concatenated [groups, 64, 8], of all 'V' blocks concatenated, and the permutation that puts the produced words in block order."""
    key = (id(blocks), str(device))
    plan = _plans.get(key)
    if plan is None:
        ia = [idx.reshape(-1, 64, 8) for kind, idx in blocks if kind == "A"]
        iv = [idx.reshape(-1) for kind, idx in blocks if kind != "A"]
        ia = torch.cat(ia) if ia else torch.zeros(0, 64, 8, dtype=torch.int64)
        iv = torch.cat(iv) if iv else torch.zeros(0, dtype=torch.int64)
        n_a, pa, pv, perm = ia.shape[0] * 3 * 64 * 4, 0, 0, []
        for kind, idx in blocks:
            if kind == "A":
                n = idx.numel() // 8 * 3 * 4
                perm.append(torch.arange(pa, pa + n))
                pa += n
            else:
                perm.append(n_a + torch.arange(pv, pv + idx.numel()))
                pv += idx.numel()
        plan = _plans[key] = (blocks, ia.to(device), iv.to(device), torch.cat(perm).to(device))
    return plan[1:]


def pack_blocks(flat, blocks):
    """Gather every block from the flat parameter vector; 'A' blocks become [group][piece][lane][8 bf16] (viewed as
    float32 words), 'V' blocks stay fp32.  Layout: csrc/mlp_common.hpp.  All blocks of a kind go through one gather."""
    ia, iv, perm = _plan(blocks, flat.device)
    if flat.is_cuda and not flat.requires_grad and flat.dtype == torch.float32 and not torch.is_grad_enabled():
        # the whole plan in ONE launch (csrc/map_tail.hip::k_pack_blocks) -- what the fused engine always asks for (detached packs)
        from .._native import lib, check
        flat = flat.contiguous()
        out = torch.empty(perm.numel(), device=flat.device, dtype=torch.float32)
        check(lib.nsa_pack_blocks(flat.data_ptr(), ia.data_ptr(), ia.numel(), iv.data_ptr(), iv.numel(), perm.data_ptr(),
                                  perm.numel(), out.data_ptr(), torch.cuda.current_stream().cuda_stream))
        return out
    words = torch.stack(split_pieces(flat[ia]), 1).contiguous().view(torch.float32).reshape(-1)
    return torch.cat([words, flat[iv]])[perm]


@functools.lru_cache(maxsize=None)
def sdf_net_index(NH, L, C):
    """Index map of one SDF network's packed block (order = csrc/sdf_net.hpp::SdfPack<NH>) into
    flat = cat[lin0.W(64x71), lin0.b, lin1.W, lin1.b, ..., lin{NH}.W(65x64), lin{NH}.b]."""
    n_in = 39 + L * C
    assert n_in == 71
    offs, o = [], 0
    shapes = [(64, n_in)] + [(64, 64)] * (NH - 1) + [(65, 64)]
    for (a, b) in shapes:
        offs.append((o, o + a * b))   # (W offset, bias offset)
        o += a * b + a
    Wo = lambda k: offs[k][0]
    Bo = lambda k: offs[k][1]
    n_cols = lambda k: shapes[k][1]
    blocks = []
    # W0, B0
    blocks.append(a_block(2, SDF_IN_STEPS, lambda mt, i, s, h: (
        Wo(0) + (32 * mt + i) * n_in + f if (f := sdf_in_feature(s, h, L, C)) >= 0 else -1)))
    blocks.append(vec_block(2, lambda f: Bo(0) + f))
    for k in range(1, NH):
        blocks.append(a_block(2, 32, lambda mt, i, s, h, k=k: Wo(k) + (32 * mt + i) * 64 + hid_feature(s, h)))
        blocks.append(vec_block(2, lambda f, k=k: Bo(k) + f))
    # WSDF, BSDF
    blocks.append(vec_block(2, lambda f: Wo(NH) + f))
    bs = np.full(64, -1, dtype=np.int64)
    bs[0] = Bo(NH)
    blocks.append(("V", bs))
    # WFEAT, BFEAT (rows 1..64 of the last layer)
    blocks.append(a_block(2, 32, lambda mt, i, s, h: Wo(NH) + (1 + 32 * mt + i) * 64 + hid_feature(s, h)))
    blocks.append(vec_block(2, lambda f: Bo(NH) + 1 + f))
    # transposed hidden layers, k = NH-1 .. 1:  A[row = in-feature][slot = out-feature]
    for k in range(NH - 1, 0, -1):
        blocks.append(a_block(2, 32, lambda mt, i, s, h, k=k: Wo(k) + hid_feature(s, h) * 64 + (32 * mt + i)))

    def w0t(mt, i, s, h):
        q, hh = row_slot(mt, i)
        if q >= SDF_IN_STEPS:
            return -1
        f = sdf_in_feature(q, hh, L, C)
        return Wo(0) + hid_feature(s, h) * n_in + f if f >= 0 else -1
    blocks.append(a_block(3, 32, w0t))
    # WFEATT: A[row = hidden in-feature][slot = feature output (1 + ...)]
    blocks.append(a_block(2, 32, lambda mt, i, s, h: Wo(NH) + (1 + hid_feature(s, h)) * 64 + (32 * mt + i)))
    return _finish(blocks, o), o


def _finish(blocks, zero_index):
    out = []
    for kind, idx in blocks:
        idx = idx.copy()
        idx[idx < 0] = zero_index            # index of the appended zero
        out.append((kind, torch.from_numpy(idx)))
    return tuple(out)


def a_floats(MT, KS):
    return MT * ((KS + 7) // 8) * 3 * 64 * 4


def sdf_pack_size(NH):
    hh = a_floats(2, 32)
    return (a_floats(2, SDF_IN_STEPS) + 64 + (NH - 1) * (hh + 64) + 64 + 64 + hh + 64 + (NH - 1) * hh
            + a_floats(3, 32) + hh)


def effective_weight(lin):
    """Weight of a (possibly legacy weight-normed) nn.Linear, recomputed from the live parameters."""
    if hasattr(lin, "weight_g"):
        return torch._weight_norm(lin.weight_v, lin.weight_g, 0)
    return lin.weight


class FlatWeightNorm(torch.autograd.Function):
    """(weight_v_0, weight_g_0, bias_0, weight_v_1, ..) -> flat effective parameter vector, forward and backward one launch each
    (csrc/map_tail.hip::k_weight_norm_flat[_bwd]): what per-layer torch._weight_norm + reshape + cat compute, with ATen's formulas.
    A mapping iteration re-derives three networks' weights every step (base_networks.py:137-141 registers weight_norm on
    every Linear), which as torch ops was ~27 launches per iteration."""

    @staticmethod
    def _desc(params):
        from .._native import WnLayer
        n = len(params) // 3
        arr = (WnLayer * n)()
        for l in range(n):
            v, g, b = params[3 * l:3 * l + 3]
            arr[l] = WnLayer(v.data_ptr(), g.data_ptr(), b.data_ptr(), v.shape[0], v.shape[1])
        return arr, n

    @staticmethod
    def forward(ctx, *params):
        from .._native import lib, check
        params = tuple(p.detach().contiguous() for p in params)
        arr, n = FlatWeightNorm._desc(params)
        rows = sum(int(params[3 * l].shape[0]) for l in range(n))
        size = sum(int(params[3 * l].numel()) + int(params[3 * l].shape[0]) for l in range(n))
        dev = params[0].device
        flat = torch.empty(size + 1, device=dev)
        norms = torch.empty(rows, device=dev)
        with torch.cuda.device(dev):
            check(lib.nsa_weight_norm_flat(arr, n, flat.data_ptr(), norms.data_ptr(), torch.cuda.current_stream().cuda_stream))
        ctx.save_for_backward(norms, *params)
        return flat

    @staticmethod
    def backward(ctx, g_flat):
        from .._native import lib, check
        norms, *params = ctx.saved_tensors
        arr, n = FlatWeightNorm._desc(params)
        g_flat = g_flat.contiguous()
        out = torch.empty(sum(p.numel() for p in params), device=g_flat.device)
        with torch.cuda.device(g_flat.device):
            check(lib.nsa_weight_norm_flat_backward(arr, n, norms.data_ptr(), g_flat.data_ptr(), out.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream))
        grads, o = [], 0
        for l in range(n):
            for i, p in enumerate(params[3 * l:3 * l + 3]):
                grads.append(out[o:o + p.numel()].view(p.shape) if ctx.needs_input_grad[3 * l + i] else None)
                o += p.numel()
        return tuple(grads)


def _wn_params(net):
    """[weight_v, weight_g, bias] per layer when every Linear of ``net`` is a weight-normed float32 CUDA layer, else None."""
    out = []
    for l in range(net.num_layers - 1):
        lin = getattr(net, "lin" + str(l))
        if not (hasattr(lin, "weight_g") and hasattr(lin, "weight_v") and lin.bias is not None):
            return None
        ps = [lin.weight_v, lin.weight_g, lin.bias]
        if not all(p.is_cuda and p.dtype == torch.float32 for p in ps) or lin.weight_v.dim() != 2:
            return None
        out += ps
    return out if 3 <= len(out) <= 24 else None


def flat_params(net):
    """Flat effective parameter vector [W_0, b_0, W_1, b_1, .., 0] of an MLP (differentiable)."""
    wn = _wn_params(net)
    if wn is not None:
        if torch.is_grad_enabled():
            return FlatWeightNorm.apply(*wn)
        # detached (the packs of every tiling of this network ask for it): once per parameter version
        key = tuple((p.data_ptr(), p._version) for p in wn)
        hit = net.__dict__.get("_flat_detached")
        if hit is None or hit[0] != key:
            hit = net.__dict__["_flat_detached"] = (key, FlatWeightNorm.apply(*wn))
        return hit[1]
    parts = []
    for l in range(net.num_layers - 1):
        lin = getattr(net, "lin" + str(l))
        parts += [effective_weight(lin).reshape(-1), lin.bias.reshape(-1)]
    parts.append(parts[0].new_zeros(1))
    return torch.cat(parts)


def pack_sdf_net(net):
    """ImplicitNetworkGrid -> packed float32 device tensor (differentiable gather of the effective parameters)."""
    NH = net.num_layers - 2
    enc = net.encoding
    blocks, n = sdf_net_index(NH, enc.num_levels, enc.level_dim)
    flat = flat_params(net)
    assert flat.numel() == n + 1, (flat.numel(), n)
    packed = pack_blocks(flat, blocks)
    assert packed.numel() == sdf_pack_size(NH)
    return packed


# ------------------------------------------------------------------------------------------------ colour network
def col_in_feature(s, h):
    """reference column of rendering_input (base_networks.py:346: [x, PE4(view), normals, features, grid]) of
    first-layer slot (s, h); table in csrc/render_colour.hip."""
    if s < 32:
        return 33 + hid_feature(s, h)
    if s == 32:
        return 0 if h == 0 else 1
    if s == 33:
        return 2 if h == 0 else 3
    if s == 34:
        return 4 if h == 0 else 5
    if s == 35:
        return 30 if h == 0 else 31
    if s == 36:
        return 32 if h == 0 else -1
    if s < 49:
        j, which = divmod(s - 37, 2)
        g = 2 * j + h
        k, dd = divmod(g, 3)
        return 6 + 6 * k + dd + (3 if which else 0)
    jl, c = divmod(s - 49, 2)
    return 97 + (2 * jl + h) * 2 + c


@functools.lru_cache(maxsize=None)
def colour_net_index():
    """Index map of the colour network's packed block (order = csrc/render_colour.hip::ColPack) into
    flat = cat[lin0.W(64x129), lin0.b, lin1.W(64x64), lin1.b, lin2.W(3x64), lin2.b]."""
    n_in = 129
    W0o, B0o = 0, 64 * n_in
    W1o = B0o + 64
    B1o = W1o + 64 * 64
    W2o = B1o + 64
    B2o = W2o + 3 * 64
    total = B2o + 3
    blocks = [
        a_block(2, COL_IN_STEPS, lambda mt, i, s, h: (
            W0o + (32 * mt + i) * n_in + f if (f := col_in_feature(s, h)) >= 0 else -1)),
        vec_block(2, lambda f: B0o + f),
        a_block(2, 32, lambda mt, i, s, h: W1o + (32 * mt + i) * 64 + hid_feature(s, h)),
        vec_block(2, lambda f: B1o + f),
    ]
    for j in range(3):
        blocks.append(vec_block(2, lambda f, j=j: W2o + j * 64 + f))
    b2 = np.full(64, -1, dtype=np.int64)
    b2[:3] = [B2o, B2o + 1, B2o + 2]
    blocks.append(("V", b2))
    blocks.append(a_block(2, 32, lambda mt, i, s, h: W1o + hid_feature(s, h) * 64 + (32 * mt + i)))

    def w0t(mt, i, s, h):
        q, hh = row_slot(mt, i)
        if q >= COL_IN_STEPS:
            return -1
        f = col_in_feature(q, hh)
        return W0o + hid_feature(s, h) * n_in + f if f >= 0 else -1
    blocks.append(a_block(5, 32, w0t))
    return _finish(blocks, total), total


COL_PACK_SIZE = a_floats(2, COL_IN_STEPS) + 64 + a_floats(2, 32) + 64 + 192 + 64 + a_floats(2, 32) + a_floats(5, 32)


def pack_colour_net(net):
    """RenderingNetwork (mode idr, 129->64->64->3) -> packed float32 device tensor."""
    blocks, n = colour_net_index()
    flat = flat_params(net)
    assert flat.numel() == n + 1, (flat.numel(), n)
    packed = pack_blocks(flat, blocks)
    assert packed.numel() == COL_PACK_SIZE
    return packed


# ====================================================================================================================
# "Quad" layout of the SDF-network kernels (csrc/mlp16.hpp, csrc/sdf_net4.hpp): v_mfma_f32_16x16x32_bf16, a wave = 16 points,
# FOUR lanes per point (lane = point j + 16 * quarter q).  A 64-feature activation is 16 floats per lane, index s:
#   feature(s, q) = 16 * (s >> 2) + 4 * q + (s & 3)          (= row 4q + r of output tile t = s >> 2, r = s & 3)
# and k-group g of the next GEMM (32 k-values, 8 per lane) takes act[8g .. 8g+7] of every lane.
QIN_STEPS = 24          # first-layer slots per lane: 3 k-groups (positional encoding 2 groups incl. x, grid 1 group)


def qfeat(s, q):
    """hidden feature held at activation index s of quarter-lane q"""
    return 16 * (s >> 2) + 4 * q + (s & 3)


def sdf_in_feature4(s, q, C=8):
    """reference input-feature index (base_networks.py:155-164: [x, PE6(x), grid(32)]) of first-layer slot s of quarter q, or
    -1 (zero pad); the table of csrc/sdf_net4.hpp.  sin / cos of 2^k x_d are features 3+6k+d / 6+6k+d."""
    pe = lambda k, d, which: 3 + 6 * k + d + (3 if which else 0)
    if s < 6:                                  # pairs n = 0..2: frequency q, coordinate n
        return pe(q, s >> 1, s & 1)
    if s < 8:                                  # pair 3: frequency 4 + (q & 1), coordinate 0 or 2
        return pe(4 + (q & 1), 2 if (q & 2) else 0, s & 1)
    if s < 10:                                 # pair 4: q < 2: frequency 4 + q, coordinate 1;  q == 3: x_0, x_1
        if q < 2:
            return pe(4 + q, 1, s & 1)
        return (s - 8) if q == 3 else -1
    if s == 10:
        return 2 if q == 3 else -1
    if s < 16:
        return -1
    jl, c = divmod(s - 16, C)
    return 39 + (q + 4 * jl) * C + c


def a_block16(MT, KG, elem):
    """int64 index block [MT][KG][64 lanes][8]; elem(mt, i, g, kq, e) -> flat parameter index or -1 (zero): the weight that
    lane (i = lane & 15, kq = lane >> 4) contributes as A[16 mt + i][k = (g, kq, e)]."""
    out = np.full((MT, KG, 64, 8), -1, dtype=np.int64)
    for mt in range(MT):
        for g in range(KG):
            for lane in range(64):
                for e in range(8):
                    out[mt, g, lane, e] = elem(mt, lane & 15, g, lane >> 4, e)
    return ("A", out.reshape(-1))


def vec_block16(elem):
    """activation-layout vector [q * 16 + s]; elem(feature) -> flat index."""
    out = np.full(64, -1, dtype=np.int64)
    for q in range(4):
        for s in range(16):
            out[q * 16 + s] = elem(qfeat(s, q))
    return ("V", out)


@functools.lru_cache(maxsize=None)
def sdf_net_index4(NH, C):
    """Index map of one SDF network's packed block in the quad layout (order = csrc/sdf_net4.hpp::SdfPack4<NH>); C = grid
    channels per level (8: one level per quarter-lane, 4: two)."""
    n_in = 71
    offs, o = [], 0
    shapes = [(64, n_in)] + [(64, 64)] * (NH - 1) + [(65, 64)]
    for (a, b) in shapes:
        offs.append((o, o + a * b))
        o += a * b + a
    Wo = lambda k: offs[k][0]
    Bo = lambda k: offs[k][1]
    hk = lambda g, kq, e: qfeat(8 * g + e, kq)            # hidden feature supplied as k-value (g, kq, e)
    blocks = []
    blocks.append(a_block16(4, 3, lambda mt, i, g, kq, e: (
        Wo(0) + (16 * mt + i) * n_in + f if (f := sdf_in_feature4(8 * g + e, kq, C)) >= 0 else -1)))
    blocks.append(vec_block16(lambda f: Bo(0) + f))
    for k in range(1, NH):
        blocks.append(a_block16(4, 2, lambda mt, i, g, kq, e, k=k: Wo(k) + (16 * mt + i) * 64 + hk(g, kq, e)))
        blocks.append(vec_block16(lambda f, k=k: Bo(k) + f))
    blocks.append(vec_block16(lambda f: Wo(NH) + f))                       # sdf row
    bs = np.full(64, -1, dtype=np.int64)
    bs[0] = Bo(NH)
    blocks.append(("V", bs))
    blocks.append(a_block16(4, 2, lambda mt, i, g, kq, e: Wo(NH) + (1 + 16 * mt + i) * 64 + hk(g, kq, e)))   # feature rows
    blocks.append(vec_block16(lambda f: Bo(NH) + 1 + f))
    for k in range(NH - 1, 0, -1):                                         # transposed hidden layers
        blocks.append(a_block16(4, 2, lambda mt, i, g, kq, e, k=k: Wo(k) + hk(g, kq, e) * 64 + (16 * mt + i)))

    def w0t(mt, i, g, kq, e):                                              # rows = first-layer slots: tile mt, row i = 4 q + r
        f = sdf_in_feature4(4 * mt + (i & 3), i >> 2, C)
        return Wo(0) + hk(g, kq, e) * n_in + f if f >= 0 else -1
    blocks.append(a_block16(6, 2, w0t))
    blocks.append(a_block16(4, 2, lambda mt, i, g, kq, e: Wo(NH) + (1 + hk(g, kq, e)) * 64 + (16 * mt + i)))  # feature rows^T
    return _finish(blocks, o), o


def a_floats16(MT, KG):
    return MT * KG * 3 * 64 * 4


def sdf_pack_size4(NH):
    hh = a_floats16(4, 2)
    return (a_floats16(4, 3) + 64 + (NH - 1) * (hh + 64) + 64 + 64 + hh + 64 + (NH - 1) * hh + a_floats16(6, 2) + hh)


def pack_sdf_net4(net):
    """ImplicitNetworkGrid -> packed float32 device tensor in the quad layout."""
    NH = net.num_layers - 2
    enc = net.encoding
    assert enc.num_levels * enc.level_dim == 32 and enc.level_dim in (4, 8)
    blocks, n = sdf_net_index4(NH, enc.level_dim)
    flat = flat_params(net)
    assert flat.numel() == n + 1, (flat.numel(), n)
    packed = pack_blocks(flat, blocks)
    assert packed.numel() == sdf_pack_size4(NH)
    return packed
