"""The two operand forms of the library's fp32 GEMMs (csrc/mlp_common.hpp::NSA_FORM), emulated bit by bit in numpy and held to float64:

  form 3  three exact bf16 pieces per operand (truncation split), six products per 16-k block           (rounds 1-6a)
  form 2  two round-to-nearest fp16 pieces of  w * 2^9  resp. of  b * s  (s = the point's power of two that puts its largest |b| into
          [2^13, 2^14)), four products per block, unscaled at the end                                    (round 6b: the default)

Both accumulate exact products of a 16-k block into an fp32 accumulator, block after block and product after product, as the matrix
instruction sequence of mma_group does (the hardware's internal order inside one instruction is not modelled: one rounding per
instruction).  Claims held here, on the operand families the kernels see (Softplus activations, first-layer inputs with 1e-4 grid
features, cotangents whose per-point magnitudes span twelve decades):

  * form 2 is at least as close to float64 as form 3 (rms and max of |error| / sum |w||b|) -- it has a third fewer accumulator roundings;
  * both are closer than an fp32 multiply-add chain, the arithmetic of the reference's nn.Linear on a CPU;
  * the split of form 2 holds every operand to 2^-23 (relative), the packed weights included, down to |w| = 2^-11; smaller weights to
    2^-34 absolute.

The kernels themselves are held to the oracle by the GPU suite at unchanged tolerances (tests/test_configs_gpu.py, test_fused_gpu.py ...);
the packed pieces to their torch restatement bit for bit (tests/test_pack_gpu.py)."""
import numpy as np
import pytest

W_SCALE = 512.0


def bf16_trunc(x):
    return (x.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split_bf16x3(x):
    p0 = bf16_trunc(x)
    r1 = (x - p0).astype(np.float32)
    p1 = bf16_trunc(r1)
    return p0, p1, bf16_trunc((r1 - p1).astype(np.float32))


def f16(x):
    return x.astype(np.float16).astype(np.float32)


def split_f16x2(t):
    h0 = f16(t)
    return h0, f16((t - h0).astype(np.float32))


def mfma_chain(terms, K, kb=16):
    acc = np.zeros((terms[0][0].shape[0], terms[0][1].shape[1]), np.float32)
    for k0 in range(0, K, kb):
        for A, B in terms:
            acc = (acc.astype(np.float64) + A[:, k0:k0 + kb].astype(np.float64) @ B[k0:k0 + kb].astype(np.float64)).astype(np.float32)
    return acc


def gemm_form3(W, X):
    w, x = split_bf16x3(W), split_bf16x3(X)
    return mfma_chain([(w[i], x[j]) for i, j in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0))], W.shape[1])


def point_scale(X):
    """csrc/mlp_common.hpp::point_scale_of: 2^(140 - e), e = biased exponent of the point's largest |b| clamped to [47, 207]"""
    m = np.abs(X).max(axis=0, keepdims=True).astype(np.float32)
    e = np.clip(m.view(np.uint32) >> 23, 47, 207).astype(np.int64)
    return np.ldexp(np.float32(1.0), (140 - e).astype(np.int32)).astype(np.float32)


def gemm_form2(W, X):
    s = point_scale(X)
    w0, w1 = split_f16x2((W * np.float32(W_SCALE)).astype(np.float32))
    x0, x1 = split_f16x2((X * s).astype(np.float32))
    acc = mfma_chain([(w1, x1), (w0, x1), (w1, x0), (w0, x0)], W.shape[1])
    return (acc.astype(np.float64) / (np.float64(W_SCALE) * s.astype(np.float64))).astype(np.float32)


def fp32_chain(W, X):
    acc = np.zeros((W.shape[0], X.shape[1]), np.float32)
    for k in range(W.shape[1]):
        acc = (acc + (W[:, k:k + 1] * X[k:k + 1]).astype(np.float32)).astype(np.float32)
    return acc


def families():
    rng = np.random.default_rng(0)
    N = 2048
    W = (rng.standard_normal((64, 64)) * 0.15).astype(np.float32)
    yield "hidden layer, Softplus activations", W, (np.log1p(np.exp(rng.standard_normal((64, N)) * 2)) * 0.3).astype(np.float32)
    X0 = np.concatenate([rng.uniform(-1, 1, (3, N)), np.sin(rng.uniform(-40, 40, (36, N))), rng.uniform(-1e-4, 1e-4, (32, N)),
                         np.zeros((9, N))]).astype(np.float32)
    yield "first layer (xyz, PE, 1e-4 grid features)", (rng.standard_normal((64, 80)) * 0.1).astype(np.float32), X0
    G = (rng.standard_normal((64, N)) * 10.0 ** rng.uniform(-12, 0, (1, N))).astype(np.float32)
    yield "backward, cotangents spanning 1e-12 .. 1 per point", W.T.copy(), G


@pytest.mark.parametrize("name,W,X", list(families()), ids=lambda v: v if isinstance(v, str) else "")
def test_form2_is_no_further_from_float64_than_form3_and_an_fp32_chain(name, W, X):
    ref = W.astype(np.float64) @ X.astype(np.float64)
    unit = np.abs(W).astype(np.float64) @ np.abs(X).astype(np.float64)
    unit[unit == 0] = 1.0
    err = {}
    for nm, fn in (("fp32 chain", fp32_chain), ("form 3", gemm_form3), ("form 2", gemm_form2)):
        e = np.abs(fn(W, X).astype(np.float64) - ref) / unit
        err[nm] = (float(e.max()), float(np.sqrt((e ** 2).mean())))
    print(name, {k: ("max %.2f x 2^-24" % (v[0] / 2.0 ** -24), "rms %.2e" % v[1]) for k, v in err.items()})
    assert err["form 2"][1] <= 1.02 * err["form 3"][1] and err["form 2"][0] <= 1.1 * err["form 3"][0], err
    assert err["form 2"][1] <= err["fp32 chain"][1] and err["form 3"][1] <= err["fp32 chain"][1], err
    assert err["form 2"][0] < 4 * 2.0 ** -24, err            # (measured 2.1-2.7 x 2^-24 of sum |w||b|)


def test_the_two_piece_split_holds_every_operand_to_2_pow_minus_23():
    rng = np.random.default_rng(1)
    # activations: any magnitude (the exponent clamp of point_scale_of is +-80, i.e. 1e-24 .. 1e24), scaled by the point's power of two
    X = (rng.standard_normal((64, 4096)) * 10.0 ** rng.uniform(-22, 22, (1, 4096))).astype(np.float32)
    s = point_scale(X)
    t = (X * s).astype(np.float32)
    assert np.abs(t).max() < 2.0 ** 14 and np.isfinite(t).all()
    h0, h1 = split_f16x2(t)
    big = np.abs(t) >= 0.25                                   # (h1 is a normal fp16 number from here up)
    rel = np.abs((h0.astype(np.float64) + h1) - t)[big] / np.abs(t[big])
    assert rel.max() <= 2.0 ** -23, rel.max()
    assert np.abs((h0.astype(np.float64) + h1) - t)[~big].max() <= 2.0 ** -25     # below: fp16's subnormal spacing / 2, 2^-38 of the largest
    # weights: w * 2^9, round-to-nearest twice (fused/pack.py::split_f16x2, csrc/map_tail.hip::h2_piece)
    w = (rng.standard_normal(200000) * 10.0 ** rng.uniform(-6, 1.5, 200000)).astype(np.float32)
    w = w[np.abs(w) < 127.9]
    t = (w * np.float32(W_SCALE)).astype(np.float32)
    h0, h1 = split_f16x2(t)
    err = np.abs((h0.astype(np.float64) + h1) - t)
    full = np.abs(w) >= 2.0 ** -11                          # (512 w >= 0.25: h1 is a normal fp16 number or exactly representable)
    assert (err[full] / np.abs(t[full])).max() <= 2.0 ** -23
    assert err[~full].max() / W_SCALE <= 2.0 ** -34
