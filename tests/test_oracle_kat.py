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
