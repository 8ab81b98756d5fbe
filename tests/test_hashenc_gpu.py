"""Parity of the HIP hash-grid operator (through the C ABI: ctypes -> libnicer_slam_amd.so) with the CPU oracle
and the reference-captured goldens.  Needs an MI355X.

Tolerances (fp32): forward/Jacobian abs 1e-5 + rel 1e-4 of the table magnitude; scattered table
gradients rel 1e-3 (float atomics reorder the sums; hipcc contracts a*b+c to FMA, the oracle is unfused)."""
import numpy as np
import pytest
import torch

from helpers import load, tt, assert_close

pytestmark = pytest.mark.gpu


def _spec(L, C, base, end, logmap):
    from oracle import render_ref as R
    return R.make_grid_spec(L, C, base, end, logmap)


def _gpu_encode_suite(spec, emb, x01, v, q, r):
    """forward, J^T v, table scatter, and the second-backward products on the GPU via the product's Functions."""
    from nicer_slam_amd.hashencoder.hashgrid import hash_encode
    dev = "cuda"
    emb_g = emb.to(dev).requires_grad_(True)
    x_g = x01.to(dev).requires_grad_(True)
    v_g = v.to(dev).requires_grad_(True)
    off = spec.offsets.to(dev)
    y = hash_encode(x_g, emb_g, off, spec.per_level_scale, spec.base_resolution, True)
    (gx,) = torch.autograd.grad(y, x_g, v_g, create_graph=True)
    first = torch.autograd.grad(y, emb_g, v_g, retain_graph=True)[0]
    ((gx * q.to(dev)).sum() + (y * r.to(dev)).sum()).backward()
    return dict(y=y, gx=gx, first=first, emb_grad=emb_g.grad, v_grad=v_g.grad, x_grad=x_g.grad)


def _cpu_encode_suite(spec, emb, x01, v, q, r):
    from oracle import render_ref as R
    emb_c = emb.clone().requires_grad_(True)
    x_c = x01.clone().requires_grad_(True)
    v_c = v.clone().requires_grad_(True)
    y = R._Encode.apply(x_c, emb_c, spec.offsets, spec.per_level_scale, spec.base_resolution, True)
    (gx,) = torch.autograd.grad(y, x_c, v_c, create_graph=True)
    first = torch.autograd.grad(y, emb_c, v_c, retain_graph=True)[0]
    ((gx * q).sum() + (y * r).sum()).backward()
    return dict(y=y, gx=gx, first=first, emb_grad=emb_c.grad, v_grad=v_c.grad, x_grad=x_c.grad)


def _compare(g, c, scale=1.0):
    assert_close(g["y"], c["y"], 1e-5 * scale, 1e-4, "features")
    jac_scale = float(c["gx"].abs().max()) + 1e-12
    assert_close(g["gx"], c["gx"], 2e-5 * jac_scale, 1e-4, "J^T v")
    assert_close(g["v_grad"], c["v_grad"], 2e-5 * float(c["v_grad"].abs().max() + 1e-12), 1e-4, "d/dv (J q)")
    for k in ("first", "emb_grad"):
        assert_close(g[k], c[k], 1e-4 * float(c[k].abs().max() + 1e-12), 1e-3, k)
    if c["x_grad"] is not None:
        assert_close(g["x_grad"], c["x_grad"], 2e-5 * float(c["x_grad"].abs().max() + 1e-12), 1e-4, "x.grad")


@pytest.mark.parametrize("name", ["enc_coarse", "enc_fine", "enc_colour"])
def test_golden_function_level(name):
    """GPU vs vectors captured through the reference's own wrappers (incl. x=0, x=1, out-of-range, duplicates)."""
    from nicer_slam_amd.hashencoder.hashgrid import HashEncoder
    fx = load(name)
    L, C, base, end, logmap = [int(t) for t in fx["meta_grid"]]
    enc = HashEncoder(num_levels=L, level_dim=C, base_resolution=base, desired_resolution=end,
                      log2_hashmap_size=logmap).cuda()
    assert enc.offsets.cpu().tolist() == fx["param_offsets"].tolist()
    enc.embeddings.data = tt(fx["param_embeddings"]).cuda()
    x = tt(fx["in_x"]).cuda().requires_grad_(True)
    v = tt(fx["in_v"]).cuda().requires_grad_(True)
    y = enc(x)
    (gx,) = torch.autograd.grad(y, x, v, create_graph=True)
    first = torch.autograd.grad(y, enc.embeddings, v, retain_graph=True)[0]
    ((gx * tt(fx["in_q"]).cuda()).sum() + (y * tt(fx["in_r"]).cuda()).sum()).backward()
    g = dict(y=y, gx=gx, first=first, emb_grad=enc.embeddings.grad, v_grad=v.grad, x_grad=x.grad)
    c = dict(y=tt(fx["out_y"]), gx=tt(fx["out_gx"]), first=tt(fx["out_first_emb"]), emb_grad=tt(fx["out_emb_grad"]),
             v_grad=tt(fx["out_v_grad"]), x_grad=tt(fx["out_x_grad"]))
    _compare(g, c)
    assert float(y[3].abs().max()) == 0 and float(y[4].abs().max()) == 0   # out-of-range rows are exactly zero


@pytest.mark.parametrize("name", ["twin_dense", "twin_dense_c8", "twin_dense_c2"])
def test_reference_twin_second_order(name):
    """HIP operator vs stock autograd through the reference's pure-torch twin (HashEncoder.torch_forward,
    hashgrid.py:217-299): value, J^T v, table scatter and the two second-order products the CUDA path keeps
    (hashencoder.cu:405-625) -- expected values that never touched the C restatement."""
    from nicer_slam_amd.hashencoder.hashgrid import HashEncoder
    from test_oracle_golden import twin_checks
    fx = load(name)
    L, C, base, end, logmap = [int(t) for t in fx["meta_grid"]]
    enc = HashEncoder(num_levels=L, level_dim=C, base_resolution=base, desired_resolution=end,
                      log2_hashmap_size=logmap).cuda()
    assert enc.offsets.cpu().tolist() == fx["param_offsets"].tolist()
    enc.embeddings.data = tt(fx["param_embeddings"]).cuda()
    x = tt(fx["in_x"]).cuda().requires_grad_(True)
    v = tt(fx["in_v"]).cuda().requires_grad_(True)
    y = enc(x)
    (gx,) = torch.autograd.grad(y, x, v, create_graph=True)
    (first,) = torch.autograd.grad(y, enc.embeddings, v, retain_graph=True)
    v2, e2 = torch.autograd.grad((gx * tt(fx["in_q"]).cuda()).sum(), [v, enc.embeddings])
    assert_close(y, fx["out_y"], 5e-5, 1e-4, "y vs torch_forward")
    assert_close(gx, fx["out_gx"], 5e-5 * float(np.abs(fx["out_gx"]).max()), 1e-4, "J^T v vs autograd(torch_forward)")
    twin_checks(fx, first, v2, e2)


def test_real_colour_geometry_sparse_table():
    """Shipped colour grid (16x2, 16->2048, 2^24 rows/level; 1 GiB) incl. the uint32 stride wrap at res 2048."""
    from nicer_slam_amd.hashencoder.hashgrid import hash_encode
    fx = load("enc_colour_real_sparse")
    L, C, base, end, logmap = [int(t) for t in fx["meta_grid"]]
    spec = _spec(L, C, base, end, logmap)
    emb = torch.zeros(spec.n_rows, C, device="cuda")
    emb[tt(fx["param_rows"]).cuda()] = tt(fx["param_vals"]).cuda()
    x01 = tt(fx["in_x01"]).cuda().requires_grad_(True)
    y = hash_encode(x01, emb, spec.offsets.cuda(), spec.per_level_scale, base, True)
    (gx,) = torch.autograd.grad(y, x01, tt(fx["in_v"]).cuda())
    assert_close(y, fx["out_y"], 1e-5, 1e-4, "y")
    assert_close(gx, fx["out_gx"], 2e-5 * float(np.abs(fx["out_gx"]).max()), 1e-4, "gx")


GEOMS = {"coarse": (4, 8, 32, 32, 19), "fine": (8, 4, 32, 128, 19), "colour_small": (16, 2, 16, 512, 19)}


@pytest.mark.parametrize("geom", list(GEOMS))
@pytest.mark.parametrize("B", [1, 255, 20000])
def test_oracle_parity_seeded(geom, B):
    """Shipped SDF-grid geometries (and a 2^19-capped colour geometry) at ragged batch sizes vs the C oracle."""
    spec = _spec(*GEOMS[geom])
    g = torch.Generator().manual_seed(hash((geom, B)) & 0xFFFF)
    emb = (torch.rand(spec.n_rows, spec.level_dim, generator=g) * 2 - 1)
    x01 = torch.rand(B, 3, generator=g)
    if B > 8:
        x01[0] = 0.0
        x01[1] = 1.0
        x01[2, 0] = 1.5        # out of range
        x01[3] = x01[4]        # duplicate
    v = torch.randn(B, spec.out_dim, generator=g)
    q = torch.randn(B, 3, generator=g)
    r = torch.randn(B, spec.out_dim, generator=g)
    _compare(_gpu_encode_suite(spec, emb, x01, v, q, r), _cpu_encode_suite(spec, emb, x01, v, q, r))


def test_empty_batch():
    from nicer_slam_amd.hashencoder.hashgrid import hash_encode
    spec = _spec(*GEOMS["coarse"])
    emb = torch.zeros(spec.n_rows, 8, device="cuda")
    y = hash_encode(torch.zeros(0, 3, device="cuda"), emb, spec.offsets.cuda(), spec.per_level_scale, 32, False)
    assert y.shape == (0, 32)


def test_error_behaviour_matches_reference():
    """RuntimeError texts of hashencoder.cu:16-19,637: CUDA / contiguous / int offsets / C in {1,2,4,8}."""
    from nicer_slam_amd.hashencoder.backend import _backend
    dev = "cuda"
    x = torch.rand(8, 3, device=dev)
    off = torch.tensor([0, 64, 128], dtype=torch.int32, device=dev)
    out = torch.empty(2, 8, 3, device=dev)
    with pytest.raises(RuntimeError, match="C must be 1, 2, 4, or 8"):
        _backend.hash_encode_forward(x, torch.zeros(128, 3, device=dev), off, out, 8, 3, 3, 2, 0.0, 4, False,
                                     torch.empty(1, device=dev))
    with pytest.raises(RuntimeError, match="offsets must be an int tensor"):
        _backend.hash_encode_forward(x, torch.zeros(128, 2, device=dev), off.long(), out, 8, 3, 2, 2, 0.0, 4, False,
                                     torch.empty(1, device=dev))
    with pytest.raises(RuntimeError, match="inputs must be a contiguous tensor"):
        _backend.hash_encode_forward(torch.rand(3, 8, device=dev).t(), torch.zeros(128, 2, device=dev), off, out, 8, 3,
                                     2, 2, 0.0, 4, False, torch.empty(1, device=dev))
    with pytest.raises(RuntimeError, match="C must be 1, 2, 4, or 8"):   # second backward rejects C == 1 (:708-714)
        _backend.hash_encode_second_backward(torch.zeros(2, 8, 1, device=dev), x, torch.zeros(128, 1, device=dev), off, 8,
                                             3, 1, 2, 0.0, 4, True, torch.zeros(8, 6, device=dev), x,
                                             torch.zeros(2, 8, 1, device=dev), torch.zeros(128, 1, device=dev))


def test_full_size_properties_shipped_colour_grid():
    """BASELINE full size (131 072 points, the real 1 GiB colour table): size-independent properties.
      - linearity in the table: enc(a*T1 + T2) == a*enc(T1) + enc(T2)
      - partition of unity: a constant table encodes to that constant; its Jacobian is zero
      - scatter conservation: every level's table gradient sums to the sum of the incoming gradient
        (corner weights sum to 1) -- checked per level, for in-range points only."""
    from nicer_slam_amd.hashencoder.hashgrid import hash_encode
    spec = _spec(16, 2, 16, 2048, 24)
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(7)
    B = 131072
    x01 = torch.rand(B, 3, device=dev, generator=g)
    off = spec.offsets.to(dev)
    T1 = torch.rand(spec.n_rows, 2, device=dev, generator=g) - 0.5
    enc = lambda T, jac=False: hash_encode(x01, T, off, spec.per_level_scale, 16, jac)
    e1 = enc(T1)
    T2 = torch.rand(spec.n_rows, 2, device=dev, generator=g) - 0.5
    e2 = enc(T2)
    T2.mul_(1.0).add_(T1, alpha=0.75)      # T2 <- 0.75*T1 + T2 (in place: keeps peak memory at 2 tables)
    assert_close(enc(T2), 0.75 * e1 + e2, 2e-6, 1e-5, "linearity")
    del T2
    T1.fill_(0.3125)
    xg = x01.clone().requires_grad_(True)
    yc = hash_encode(xg, T1, off, spec.per_level_scale, 16, True)
    assert_close(yc, torch.full_like(yc, 0.3125), 1e-6, 0, "partition of unity")
    (gx,) = torch.autograd.grad(yc.sum(), xg)
    assert float(gx.abs().max()) < 1e-2      # Jacobian rows are differences of equal corners (scale up to 2047)
    T1.requires_grad_(True)
    y = hash_encode(x01, T1, off, spec.per_level_scale, 16, False)
    up = torch.randn(B, 32, device=dev, generator=g)
    (gT,) = torch.autograd.grad(y, T1, up)
    for lv in range(16):
        lo, hi = int(spec.offsets[lv]), int(spec.offsets[lv + 1])
        got = gT[lo:hi].double().sum(0)
        want = up[:, 2 * lv:2 * lv + 2].double().sum(0)
        assert_close(got, want, 5e-2, 1e-3, f"scatter conservation level {lv}")
