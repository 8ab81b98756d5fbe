"""Known-answer tests pinning the oracle's index rule (hashencoder.cu:35-73)."""
import numpy as np
import pytest

from oracle import hashenc

# (cell, uint32 hash, rows, row) -- computed by hand from x*1 ^ y*2654435761 ^ z*805459861 (uint32 wrap)
KAT = [
    ((1, 2, 3), 2892625372, 2 ** 19, 128476),
    ((87, 0, 0), 87, 2 ** 19, 87),
    ((100, 200, 300), 3655970992, 2 ** 19, 110768),
    ((127, 127, 127), 2896964699, 2 ** 19, 273499),
    ((128, 128, 128), 446108288, 2 ** 19, 463488),
    ((2047, 2047, 2047), 4281096667, 2 ** 24, 2906587),
    ((295, 1, 0), 2654435478, 2 ** 24, 3635350),
    ((1000, 2000, 50), 1897115170, 2 ** 24, 1289762),
]


@pytest.mark.parametrize("cell,h,rows,row", KAT)
def test_fast_hash_known_answers(cell, h, rows, row):
    assert hashenc.fast_hash(cell) == h
    assert h % rows == row
    # independent numpy evaluation of the same formula
    primes = np.array([1, 2654435761, 805459861], dtype=np.uint64)
    v = np.uint64(0)
    for c, p in zip(cell, primes):
        v ^= (np.uint64(c) * p) & np.uint64(0xFFFFFFFF)
    assert int(v) == h


def test_dense_vs_hash_switch():
    # dense when res^3 <= rows (stride uses `res`, hashencoder.cu:60-63), hashed otherwise
    assert hashenc.level_row(2 ** 19, 71, (3, 4, 5)) == 3 + 4 * 71 + 5 * 71 * 71
    assert hashenc.level_row(2 ** 19, 87, (1, 2, 3)) == 128476          # 87^3 > 2^19 -> hash
    assert hashenc.level_row(32768, 32, (31, 31, 31)) == 32767
    # one past the last dense row wraps by the modulo (point x=1.0 reads corner+1 with weight 0)
    assert hashenc.level_row(32768, 32, (32, 31, 31)) == (32 + 31 * 32 + 31 * 1024) % 32768


def test_uint32_stride_wrap_at_2048():
    """resolution 2048, D=3: stride 2048^3 = 2^33 wraps to 0 in uint32, so `stride > rows` is false and
    the level takes the DENSE formula with a wrapped index (hashencoder.cu:56-70) -- not the hash."""
    rows = 2 ** 24
    cell = (2047, 2047, 2047)
    dense = (2047 + 2047 * 2048 + 2047 * 2048 * 2048) % 2 ** 32 % rows
    assert hashenc.level_row(rows, 2048, cell) == dense
    assert dense != hashenc.fast_hash(cell) % rows
    assert hashenc.level_row(rows, 1483, (1000, 2000, 50)) == 1289762  # 1483^3 < 2^32 -> hashed


GRIDS = {  # shipped geometries: L, C, base, end, logmap -> expected resolutions
    "coarse": (4, 8, 32, 32, 19, [32, 32, 32, 32]),
    "fine": (8, 4, 32, 128, 19, [32, 40, 48, 58, 71, 87, 106, 128]),
    "colour": (16, 2, 16, 2048, 24, [16, 23, 31, 43, 59, 81, 112, 154, 213, 295, 407, 562, 777, 1073, 1483, 2048]),
}


@pytest.mark.parametrize("name", list(GRIDS))
def test_level_geometry_matches_python_layout(name):
    """Kernel-side resolution (float32 exp2f, hashencoder.cu:180-181) agrees with the Python-side layout
    (hashgrid.py:163-168) at the shipped configs, and no level sits near a ceil() boundary."""
    from oracle import render_ref as R
    L, C, base, end, logmap, expect = GRIDS[name]
    spec = R.make_grid_spec(L, C, base, end, logmap)
    S = np.log2(spec.per_level_scale)
    for lv in range(L):
        row0, rows, res, scale = hashenc.level_geometry(spec.offsets.numpy(), lv, S, base)
        assert res == expect[lv]
        assert rows == min(2 ** logmap, res ** 3)
        frac = scale - np.floor(scale)
        assert frac == 0.0 or 1e-3 < frac < 1 - 1e-3, (lv, scale)
    assert spec.n_rows == {"coarse": 131072, "fine": 2333247, "colour": 133023682}[name]


# ---------------------------------------------------------------------------------------------------------------------------
# An independent restatement of kernel_grid (hashencoder.cu:131-283) in pure Python: arbitrary-precision ints masked to 32 bits
# for the index rule (:35-73), numpy float32 scalars for the interpolation.  It shares no code with oracle/hashenc_oracle.c and
# pins the C oracle on HASHED levels (and on the wrapped dense level) beyond the hand-computed vectors above: a wrong index picks
# a different random table row, so any disagreement is O(1), not a rounding matter.
M32 = 0xFFFFFFFF
PRIMES = (1, 2654435761, 805459861)


def _py_index(cell, rows, res):
    stride, index, d = 1, 0, 0
    while d < 3 and stride <= rows:                      # get_grid_index :56-62, uint32 arithmetic
        index = (index + cell[d] * stride) & M32
        stride = (stride * res) & M32
        d += 1
    if stride > rows:                                    # :65-67
        index = 0
        for c, p in zip(cell, PRIMES):
            index ^= (c * p) & M32
    return index % rows


def _py_level(x, level, offsets, S, H, table, C, hashed_seen):
    f32 = np.float32
    rows = int(offsets[level + 1] - offsets[level])
    # :180  exp2f of the float32 product; evaluated in float64 and rounded once (= a correctly rounded exp2f, which glibc's is to
    # within the last ulp -- numpy's own float32 exp2 is not, and one ulp of scale moves a coordinate of ~2000 cells by 1e-4)
    scale = f32(np.exp2(np.float64(f32(level) * f32(S)))) * f32(H) - f32(1.0)
    res = int(np.ceil(scale)) + 1                                                  # :181
    pos = [f32(v) * scale for v in x]
    cell = [int(np.floor(p)) for p in pos]
    t = [p - f32(c) for p, c in zip(pos, cell)]
    w = [v * v * (f32(3.0) - f32(2.0) * v) for v in t]                             # smoothstep
    dw = [f32(6.0) * v * (f32(1.0) - v) for v in t]
    stride, d = 1, 0
    while d < 3 and stride <= rows:
        stride = (stride * res) & M32
        d += 1
    hashed_seen.append(stride > rows)
    out = np.zeros(C, dtype=np.float32)
    corner_val = {}
    for idx in range(8):
        wt, q = f32(1.0), []
        for dd in range(3):
            bit = (idx >> dd) & 1
            wt = wt * (w[dd] if bit else f32(1.0) - w[dd])
            q.append(cell[dd] + bit)
        v = table[int(offsets[level]) + _py_index(q, rows, res)]
        corner_val[idx] = v
        out = out + wt * v
    jac = np.zeros((3, C), dtype=np.float32)                                       # :239-282
    for gd in range(3):
        for face in range(4):
            wt, lo, nd = scale, 0, 0
            for dd in range(3):
                if dd == gd:
                    continue
                if (face >> nd) & 1:
                    wt = wt * w[dd]
                    lo |= 1 << dd
                else:
                    wt = wt * (f32(1.0) - w[dd])
                nd += 1
            jac[gd] += wt * (corner_val[lo | (1 << gd)] - corner_val[lo]) * dw[gd]
    return out, jac


@pytest.mark.parametrize("name,L,C,base,desired,log2", [("colour-like, 2^15 rows", 16, 2, 16, 2048, 15),
                                                        ("fine SDF grid", 8, 4, 32, 128, 19)])
def test_hashed_levels_vs_pure_python_restatement(name, L, C, base, desired, log2):
    import torch
    from nicer_slam_amd.hashencoder.hashgrid import level_layout
    per_level_scale = np.exp2(np.log2(desired / base) / (L - 1))
    offsets = np.asarray(level_layout(3, L, per_level_scale, base, log2), dtype=np.int64)
    S = float(np.log2(per_level_scale))
    rng = np.random.default_rng(7)
    table = rng.uniform(-1, 1, size=(int(offsets[-1]), C)).astype(np.float32)
    B = 48
    x = rng.uniform(0.02, 0.98, size=(B, 3)).astype(np.float32)
    out = torch.zeros(L, B, C)
    dy = torch.zeros(B, L * 3 * C)
    hashenc.OracleBackend.hash_encode_forward(torch.from_numpy(x), torch.from_numpy(table), torch.from_numpy(offsets.astype(np.int32)),
                                              out, B, 3, C, L, S, base, True, dy)
    dy = dy.view(B, L, 3, C).numpy()
    hashed = []
    for b in range(B):
        for level in range(L):
            ref, jac = _py_level(x[b], level, offsets, S, base, table, C, hashed)
            np.testing.assert_allclose(out[level, b].numpy(), ref, rtol=2e-5, atol=2e-6, err_msg=f"{name}: point {b} level {level}")
            np.testing.assert_allclose(dy[b, level], jac, rtol=2e-4, atol=2e-4 * float(np.abs(jac).max() + 1), err_msg=f"{name}: J {b} {level}")
    assert any(hashed) and not all(hashed)           # the case covers hashed AND dense levels
