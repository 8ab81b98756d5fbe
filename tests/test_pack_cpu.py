"""Host-side weight packing of the fused engine vs a plain torch evaluation, through a numpy emulation of the MFMA
dataflow (tests/mfma_emu.py).  Catches slot-map / fragment-order mistakes without a GPU."""
import numpy as np
import pytest
import torch

import mfma_emu as emu
from nicer_slam_amd.fused import pack


def softplus(a):
    t = 100.0 * a
    return np.where(t > 20, a, np.log1p(np.exp(np.minimum(t, 20))) / 100.0)


def make_net(NH, L, C, seed):
    from nicer_slam_amd.model.base_networks import ImplicitNetworkGrid
    torch.manual_seed(seed)
    net = ImplicitNetworkGrid(64, 1.0, d_in=3, d_out=1, dims=[64] * NH, geometric_init=True, bias=0.6, skip_in=[],
                              weight_norm=True, multires=6, inside_outside=True, base_size=4, end_size=8, logmap=8,
                              num_levels=L, level_dim=C, divide_factor=1.0)
    g = torch.Generator().manual_seed(seed)
    for p in net.parameters():   # move away from the structured geometric init
        if p.dim() >= 1 and p.shape != net.encoding.embeddings.shape:
            p.data += 0.05 * torch.randn(p.shape, generator=g)
    return net


def slots_from_features(feats71, L, C):
    """[32 points][71] reference-ordered first-layer input -> B operand [64 lanes][36]."""
    b = np.zeros((64, pack.SDF_IN_STEPS))
    for lane in range(64):
        p, h = lane & 31, lane >> 5
        for s in range(pack.SDF_IN_STEPS):
            f = pack.sdf_in_feature(s, h, L, C)
            b[lane, s] = feats71[p, f] if f >= 0 else 0.0
    return b


@pytest.mark.parametrize("NH,L,C", [(1, 4, 8), (3, 8, 4)])
def test_sdf_net_pack_forward_and_transposes(NH, L, C):
    net = make_net(NH, L, C, seed=NH)
    packed = pack.pack_sdf_net(net).detach().float().numpy()
    assert packed.size == pack.sdf_pack_size(NH)
    rng = np.random.default_rng(0)
    h0 = rng.standard_normal((32, 71)) * 0.5
    Ws = [pack.effective_weight(getattr(net, f"lin{l}")).detach().double().numpy() for l in range(NH + 1)]
    bs = [getattr(net, f"lin{l}").bias.detach().double().numpy() for l in range(NH + 1)]
    # plain evaluation
    a = [h0 @ Ws[0].T + bs[0]]
    for k in range(1, NH):
        a.append(softplus(a[-1]) @ Ws[k].T + bs[k])
    out = softplus(a[-1]) @ Ws[NH].T + bs[NH]
    # emulated kernel dataflow (offsets as SdfPack<NH>)
    o = 0
    hh, n0, n0t = pack.a_floats(2, 32), pack.a_floats(2, 36), pack.a_floats(3, 32)
    W0 = packed[o:o + n0]; o += n0
    B0 = packed[o:o + 64]; o += 64
    WH = []
    for k in range(1, NH):
        WH.append((packed[o:o + hh], packed[o + hh:o + hh + 64])); o += hh + 64
    WSDF = packed[o:o + 64]; o += 64
    BSDF = packed[o]; o += 64
    WFEAT = packed[o:o + hh]; o += hh
    BFEAT = packed[o:o + 64]; o += 64
    WHT = {}
    for k in range(NH - 1, 0, -1):
        WHT[k] = packed[o:o + hh]; o += hh
    W0T = packed[o:o + n0t]; o += n0t
    WFEATT = packed[o:o + hh]; o += hh
    assert o == packed.size
    acc = emu.load_vec(B0, 2)
    emu.gemm_op(W0, 2, 36, slots_from_features(h0, L, C), acc)
    pre = [acc.copy()]
    for k in range(1, NH):
        nxt = emu.load_vec(WH[k - 1][1], 2)
        emu.gemm_op(WH[k - 1][0], 2, 32, emu.act_to_b(softplus(acc)), nxt)
        acc = nxt
        pre.append(acc.copy())
    act = softplus(acc)
    sdf = emu.xhalf_sum((act * emu.load_vec(WSDF, 2)).reshape(64, -1).sum(1)) + BSDF
    np.testing.assert_allclose(sdf[:32], out[:, 0], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(sdf[32:], out[:, 0], rtol=2e-6, atol=2e-6)
    feat = emu.load_vec(BFEAT, 2)
    emu.gemm_op(WFEAT, 2, 32, emu.act_to_b(act), feat)
    for lane in range(64):
        p, h = lane & 31, lane >> 5
        for t in range(2):
            for r in range(16):
                assert abs(feat[lane, t, r] - out[p, 1 + 32 * t + pack.F(r, h)]) < 2e-6
    # transposed blocks: y = W^T g for a random hidden-layout vector g
    gvec = rng.standard_normal((32, 64))
    g_b = np.zeros((64, 32))
    for lane in range(64):
        for s in range(32):
            g_b[lane, s] = gvec[lane & 31, pack.hid_feature(s, lane >> 5)]
    for k in range(1, NH):
        acc = np.zeros((64, 2, 16))
        emu.gemm_op(WHT[k], 2, 32, g_b, acc)
        want = gvec @ Ws[k]            # [32, 64 in-features]
        for lane in range(64):
            for t in range(2):
                for r in range(16):
                    assert abs(acc[lane, t, r] - want[lane & 31, 32 * t + pack.F(r, lane >> 5)]) < 2e-6
    acc = np.zeros((64, 3, 16))
    emu.gemm_op(W0T, 3, 32, g_b, acc)
    want = gvec @ Ws[0]                # [32, 71]
    for lane in range(64):
        p, h = lane & 31, lane >> 5
        for q in range(48):
            f = pack.sdf_in_feature(q, h, L, C) if q < 36 else -1
            got = acc[lane, q // 16, q % 16]
            assert abs(got - (want[p, f] if f >= 0 else 0.0)) < 2e-6, (lane, q)
    acc = np.zeros((64, 2, 16))
    emu.gemm_op(WFEATT, 2, 32, g_b, acc)
    want = gvec @ Ws[NH][1:, :]        # feature rows only
    for lane in range(64):
        for t in range(2):
            for r in range(16):
                assert abs(acc[lane, t, r] - want[lane & 31, 32 * t + pack.F(r, lane >> 5)]) < 2e-6


def test_every_input_feature_has_exactly_one_slot():
    for (L, C) in [(4, 8), (8, 4)]:
        seen = {}
        for h in range(2):
            for s in range(pack.SDF_IN_STEPS):
                f = pack.sdf_in_feature(s, h, L, C)
                if f >= 0:
                    assert f not in seen
                    seen[f] = (s, h)
        assert sorted(seen) == list(range(71))


def test_colour_net_pack():
    from nicer_slam_amd.model.base_networks import RenderingNetwork
    torch.manual_seed(5)
    net = RenderingNetwork(64, mode="idr", d_in=9, d_out=3, dims=[64, 64], weight_norm=True, multires_view=4,
                           use_grid_feature=True,
                           colour_grid=dict(base_resolution=4, desired_resolution=8, log2_hashmap_size=6))
    packed = pack.pack_colour_net(net).detach().float().numpy()
    assert packed.size == pack.COL_PACK_SIZE
    seen = {}
    for h in range(2):
        for s in range(pack.COL_IN_STEPS):
            f = pack.col_in_feature(s, h)
            if f >= 0:
                assert f not in seen
                seen[f] = (s, h)
    assert sorted(seen) == list(range(129))
    Ws = [pack.effective_weight(getattr(net, f"lin{l}")).detach().double().numpy() for l in range(3)]
    bs = [getattr(net, f"lin{l}").bias.detach().double().numpy() for l in range(3)]
    rng = np.random.default_rng(1)
    x = rng.standard_normal((32, 129))
    a1 = x @ Ws[0].T + bs[0]
    a2 = np.maximum(a1, 0) @ Ws[1].T + bs[1]
    o = np.maximum(a2, 0) @ Ws[2].T + bs[2]
    b = np.zeros((64, 65))
    for lane in range(64):
        for s in range(65):
            f = pack.col_in_feature(s, lane >> 5)
            b[lane, s] = x[lane & 31, f] if f >= 0 else 0.0
    off = 0
    hh, n0, n0t = pack.a_floats(2, 32), pack.a_floats(2, 65), pack.a_floats(5, 32)
    W0 = packed[off:off + n0]; off += n0
    B0 = packed[off:off + 64]; off += 64
    W1 = packed[off:off + hh]; off += hh
    B1 = packed[off:off + 64]; off += 64
    W2V = packed[off:off + 192]; off += 192
    B2 = packed[off:off + 64]; off += 64
    W1T = packed[off:off + hh]; off += hh
    W0T = packed[off:off + n0t]; off += n0t
    assert off == packed.size
    acc1 = emu.load_vec(B0, 2)
    emu.gemm_op(W0, 2, 65, b, acc1)
    acc2 = emu.load_vec(B1, 2)
    emu.gemm_op(W1, 2, 32, emu.act_to_b(np.maximum(acc1, 0)), acc2)
    for j in range(3):
        oj = emu.xhalf_sum((np.maximum(acc2, 0) * emu.load_vec(W2V[64 * j:64 * j + 64], 2)).reshape(64, -1).sum(1)) + B2[j]
        np.testing.assert_allclose(oj[:32], o[:, j], rtol=2e-6, atol=2e-6)
    g = rng.standard_normal((32, 64))
    g_b = np.zeros((64, 32))
    for lane in range(64):
        for s in range(32):
            g_b[lane, s] = g[lane & 31, pack.hid_feature(s, lane >> 5)]
    acc = np.zeros((64, 2, 16))
    emu.gemm_op(W1T, 2, 32, g_b, acc)
    want = g @ Ws[1]
    for lane in range(64):
        for t in range(2):
            for r in range(16):
                assert abs(acc[lane, t, r] - want[lane & 31, 32 * t + pack.F(r, lane >> 5)]) < 2e-6
    acc = np.zeros((64, 5, 16))
    emu.gemm_op(W0T, 5, 32, g_b, acc)
    want = g @ Ws[0]
    for lane in range(64):
        for q in range(80):
            f = pack.col_in_feature(q, lane >> 5) if q < 65 else -1
            assert abs(acc[lane, q // 16, q % 16] - (want[lane & 31, f] if f >= 0 else 0.0)) < 2e-6


# ---- quad layout (16-point tiles, four lanes per point; csrc/mlp16.hpp, csrc/sdf_net4.hpp) ------------------------------
def test_quad_slot_map_covers_every_input_feature_once():
    for C in (8, 4):
        seen = {}
        for q in range(4):
            for s in range(pack.QIN_STEPS):
                f = pack.sdf_in_feature4(s, q, C)
                if f >= 0:
                    assert f not in seen
                    seen[f] = (s, q)
        assert sorted(seen) == list(range(71))
    assert sorted(pack.qfeat(s, q) for q in range(4) for s in range(16)) == list(range(64))


@pytest.mark.parametrize("NH,L,C", [(1, 4, 8), (3, 8, 4)])
def test_sdf_net_quad_pack_forward_and_transposes(NH, L, C):
    net = make_net(NH, L, C, seed=10 + NH)
    packed = pack.pack_sdf_net4(net).detach().float().numpy()
    assert packed.size == pack.sdf_pack_size4(NH)
    rng = np.random.default_rng(3)
    h0 = rng.standard_normal((16, 71)) * 0.5
    Ws = [pack.effective_weight(getattr(net, f"lin{l}")).detach().double().numpy() for l in range(NH + 1)]
    bs = [getattr(net, f"lin{l}").bias.detach().double().numpy() for l in range(NH + 1)]
    a = [h0 @ Ws[0].T + bs[0]]
    for k in range(1, NH):
        a.append(softplus(a[-1]) @ Ws[k].T + bs[k])
    out = softplus(a[-1]) @ Ws[NH].T + bs[NH]
    o = 0
    hh, n0, n0t = pack.a_floats16(4, 2), pack.a_floats16(4, 3), pack.a_floats16(6, 2)
    W0 = packed[o:o + n0]; o += n0
    B0 = packed[o:o + 64]; o += 64
    WH = []
    for k in range(1, NH):
        WH.append((packed[o:o + hh], packed[o + hh:o + hh + 64])); o += hh + 64
    WSDF = packed[o:o + 64]; o += 64
    BSDF = packed[o]; o += 64
    WFEAT = packed[o:o + hh]; o += hh
    BFEAT = packed[o:o + 64]; o += 64
    WHT = {}
    for k in range(NH - 1, 0, -1):
        WHT[k] = packed[o:o + hh]; o += hh
    W0T = packed[o:o + n0t]; o += n0t
    WFEATT = packed[o:o + hh]; o += hh
    assert o == packed.size
    b0 = np.zeros((64, 24))
    for lane in range(64):
        for s in range(24):
            f = pack.sdf_in_feature4(s, lane >> 4, C)
            b0[lane, s] = h0[lane & 15, f] if f >= 0 else 0.0
    acc = emu.load_vec16(B0)
    emu.gemm16(W0, 4, 3, b0, acc)
    for k in range(1, NH):
        nxt = emu.load_vec16(WH[k - 1][1])
        emu.gemm16(WH[k - 1][0], 4, 2, softplus(acc).reshape(64, 16), nxt)
        acc = nxt
    act = softplus(acc)
    sdf = emu.quad_sum((act * emu.load_vec16(WSDF)).reshape(64, -1).sum(1)) + BSDF
    for lane in range(64):
        assert abs(sdf[lane] - out[lane & 15, 0]) < 2e-6
    feat = emu.load_vec16(BFEAT)
    emu.gemm16(WFEAT, 4, 2, act.reshape(64, 16), feat)
    for lane in range(64):
        for s in range(16):
            assert abs(feat[lane, s >> 2, s & 3] - out[lane & 15, 1 + pack.qfeat(s, lane >> 4)]) < 2e-6
    gvec = rng.standard_normal((16, 64))
    g_b = np.zeros((64, 16))
    for lane in range(64):
        for s in range(16):
            g_b[lane, s] = gvec[lane & 15, pack.qfeat(s, lane >> 4)]
    for k in range(1, NH):
        acc = np.zeros((64, 4, 4))
        emu.gemm16(WHT[k], 4, 2, g_b, acc)
        want = gvec @ Ws[k]
        for lane in range(64):
            for s in range(16):
                assert abs(acc[lane, s >> 2, s & 3] - want[lane & 15, pack.qfeat(s, lane >> 4)]) < 2e-6
    acc = np.zeros((64, 6, 4))
    emu.gemm16(W0T, 6, 2, g_b, acc)
    want = gvec @ Ws[0]
    for lane in range(64):
        for s in range(24):
            f = pack.sdf_in_feature4(s, lane >> 4, C)
            assert abs(acc[lane, s >> 2, s & 3] - (want[lane & 15, f] if f >= 0 else 0.0)) < 2e-6, (lane, s)
    acc = np.zeros((64, 4, 4))
    emu.gemm16(WFEATT, 4, 2, g_b, acc)
    want = gvec @ Ws[NH][1:, :]
    for lane in range(64):
        for s in range(16):
            assert abs(acc[lane, s >> 2, s & 3] - want[lane & 15, pack.qfeat(s, lane >> 4)]) < 2e-6
