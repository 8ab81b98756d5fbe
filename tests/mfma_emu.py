"""numpy emulation of the fused kernels' MFMA dataflow (csrc/mlp_common.hpp) -- lets the CPU suite verify the
host-side weight packing and slot maps against a plain torch MLP without a GPU.

v_mfma_f32_32x32x2_f32 semantics: D[i][j] += sum_{k<2} A[i][k] B[k][j];  lane l supplies A[l&31][l>>5] and
B[l>>5][l&31];  D[i][j] lands in lane (j + 32 hh), register r with i = (r&3) + 8(r>>2) + 4hh."""
import numpy as np

from nicer_slam_amd.fused.pack import F


def gemm_op(block, MT, KS, b, acc):
    """block: packed floats [MT][KS4][64][4]; b: [64 lanes][KS]; acc: [64][MT][16] (updated in place)."""
    KS4 = (KS + 3) // 4
    blk = np.asarray(block, dtype=np.float64).reshape(MT, KS4, 64, 4)
    for mt in range(MT):
        D = np.zeros((32, 32))
        for s in range(KS):
            a_l = blk[mt, s // 4, :, s % 4]            # per lane
            A = np.stack([a_l[:32], a_l[32:]], 1)      # [i][k]
            B = np.stack([b[:32, s], b[32:, s]], 0)    # [k][j]
            D += A @ B
        for lane in range(64):
            j, hh = lane & 31, lane >> 5
            for r in range(16):
                acc[lane, mt, r] += D[F(r, hh), j]


def load_vec(vec, MT):
    """activation-layout vector -> [64][MT][16]"""
    v = np.asarray(vec, dtype=np.float64)
    out = np.zeros((64, MT, 16))
    for lane in range(64):
        h = lane >> 5
        for mt in range(MT):
            out[lane, mt] = v[(mt * 2 + h) * 16:(mt * 2 + h) * 16 + 16]
    return out


def act_to_b(acc):
    """[64][2][16] activation -> B operand [64][32] (k-step s = 16 t + r)."""
    return acc.reshape(64, 32).copy()


def xhalf_sum(v):
    return v + np.concatenate([v[32:], v[:32]])
