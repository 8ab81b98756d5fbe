"""CPU checks of the oracle's bf16-operand emulation (oracle/render_ref.py::_LinQ): the rounding it applies is the rounding the
packed weights carry (fused/pack.py::split_bf16x3, first piece = the third slot of the two-piece form's fragment triple, split_f16x2) and the kernels' v_cvt_pk_bf16_f32 (round to nearest even), every
derivative GEMM rounds the vector it multiplies, and the fp32 path is untouched."""
import numpy as np
import torch

from oracle import render_ref as R
from nicer_slam_amd.fused.pack import split_bf16x3


def _rne_numpy(x):
    """bfloat16 round-to-nearest-even on the bit pattern (what v_cvt_pk_bf16_f32 does for finite values)"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def test_rounding_is_rne_and_equals_the_first_packed_piece():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4096, generator=g) * torch.logspace(-6, 3, 4096)
    x[:4] = torch.tensor([1.0 + 2 ** -9, 1.0 + 3 * 2 ** -9, -(1.0 + 2 ** -9), 0.0])     # ties: to even
    r = R._rne_bf16(x)
    assert np.array_equal(r.numpy().view(np.uint32), _rne_numpy(x.numpy()).view(np.uint32))
    assert torch.equal(split_bf16x3(x)[0].float(), r)
    assert float(r[0]) == 1.0 and float(r[1]) == 1.0 + 2 ** -7 and float(r[2]) == -1.0


def test_q_linear_rounds_operands_of_every_derivative_gemm():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(6, 8, generator=g, requires_grad=True)
    w = torch.randn(5, 8, generator=g, requires_grad=True)
    b = torch.randn(5, generator=g)
    rn = R._rne_bf16
    y = R.q_linear(x, w, b)
    assert torch.equal(y.detach(), rn(x.detach()) @ rn(w.detach()).t() + b)
    c = torch.randn(6, 5, generator=g)                       # cotangent with full fp32 mantissas
    (gx,) = torch.autograd.grad(y, x, c, create_graph=True)
    assert torch.equal(gx.detach(), rn(c) @ rn(w.detach()))  # first derivative GEMM: the cotangent operand is rounded
    t = torch.randn(6, 8, generator=g)
    (gc_w,) = torch.autograd.grad(gx, w, t, retain_graph=True)
    # second-order: d/dw of (rn(c) @ w) contracted with t -- the tangent operand is rounded as well
    assert torch.equal(gc_w, (rn(t).t() @ rn(c)).t())


def test_fp32_mode_is_untouched_and_bf16_modes_differ_where_they_should():
    mk = R.make_grid_spec
    torch.manual_seed(0)
    def net(prefix, n_lin, d_in, out_last):
        p = {}
        dims = [d_in] + [64] * (n_lin - 1) + [out_last]
        for l in range(n_lin):
            p[f"{prefix}.lin{l}.weight_v"] = torch.randn(dims[l + 1], dims[l]) * 0.2
            p[f"{prefix}.lin{l}.weight_g"] = torch.rand(dims[l + 1], 1) + 0.5
            p[f"{prefix}.lin{l}.bias"] = torch.randn(dims[l + 1]) * 0.1
        return p
    gc, gf, gcol = mk(4, 8, 8, 8, 10), mk(8, 4, 8, 16, 10), mk(16, 2, 4, 32, 10)
    params = {}
    params.update(net("implicit_network.coarse", 2, 71, 65))
    params.update(net("implicit_network.fine", 4, 71, 65))
    params.update(net("rendering_network", 3, 129, 3))
    for k, s in (("implicit_network.coarse", gc), ("implicit_network.fine", gf), ("rendering_network", gcol)):
        params[k + ".encoding.embeddings"] = (torch.rand(s.n_rows, s.level_dim) - 0.5) * 0.2
    x = (torch.rand(50, 3) - 0.5) * 1.2
    out = {}
    for prec in ("fp32", "bf16", "bf16_colour"):
        cfg = R.RenderConfig(coarse=R.SdfNetSpec(gc, 2), fine=R.SdfNetSpec(gf, 4), colour_grid=gcol, mlp_precision=prec)
        sdf, feat, grad = R.sdf_outputs(params, cfg, x.clone())
        rgb = R.colour_net(params, cfg, x, grad.detach(), torch.randn(50, 3, generator=torch.Generator().manual_seed(2)), feat.detach())
        out[prec] = (sdf.detach(), feat.detach(), grad.detach(), rgb.detach())
    for a, b in zip(out["fp32"][:3], out["bf16_colour"][:3]):
        assert torch.equal(a, b)                                  # fp32 SDF head: the same arithmetic
    assert not torch.equal(out["fp32"][3], out["bf16_colour"][3])
    for a, b in zip(out["fp32"], out["bf16"]):
        d = float((a - b).abs().max())
        assert 0 < d < 0.2 * float(a.abs().max()) + 0.05, d      # bf16 operands: a per-cent-level effect, not a different function
