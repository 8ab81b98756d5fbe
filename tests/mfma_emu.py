"""numpy emulation of the fused kernels' MFMA dataflow (csrc/mlp_common.hpp) -- lets the CPU suite verify the
host-side weight packing and slot maps against a plain torch MLP without a GPU.

32x32 MFMA result layout: D[i][j] lands in lane (j + 32 hh), register r with i = (r&3) + 8(r>>2) + 4hh."""
import numpy as np

from nicer_slam_amd.fused.pack import F


def _bf16_words_to_f64(words):
    """float32 words each holding two bf16 values (low half first) -> float64 array of twice the length."""
    u = np.ascontiguousarray(words, dtype=np.float32).view(np.uint32)
    lo = ((u & 0xFFFF).astype(np.uint32) << 16).view(np.float32)
    hi = (u & 0xFFFF0000).astype(np.uint32).view(np.float32)
    return np.stack([lo, hi], -1).reshape(-1).astype(np.float64)


def _f16_words_to_f64(words):
    u = np.ascontiguousarray(words, dtype=np.float32).view(np.uint32)
    lo = (u & 0xFFFF).astype(np.uint16).view(np.float16)
    hi = (u >> 16).astype(np.uint16).view(np.float16)
    return np.stack([lo, hi], -1).reshape(-1).astype(np.float64)


def block_weights(block, MT, KG):
    """packed A block [MT][KG][3 slots][64 lanes][8 halfwords] -> the fp32 weights [MT][KG][64][8] its pieces stand for, in the
    loaded library's operand form (pack.operand_form): three bf16 pieces that sum to w, or two fp16 pieces that sum to 512 w (+ the
    bf16 round-to-nearest value in the third slot, checked here)."""
    from nicer_slam_amd.fused import pack
    words = np.asarray(block, dtype=np.float32).reshape(MT, KG, 3, 64 * 4)
    if pack.operand_form() == 2:
        w = (_f16_words_to_f64(words[:, :, 0]) + _f16_words_to_f64(words[:, :, 1])).reshape(MT, KG, 64, 8) / pack.W_SCALE
        b = _bf16_words_to_f64(words[:, :, 2]).reshape(MT, KG, 64, 8)
        assert np.all(np.abs(b - w) <= 2.0 ** -8 * np.abs(w) + 1e-30), "third slot = bf16(w)"
        return w
    return _bf16_words_to_f64(words).reshape(MT, KG, 3, 64, 8).sum(2)


def gemm_op(block, MT, KS, b, acc):
    """block: packed weights [MT][KS8][3 pieces][64 lanes][8 bf16] as float32 words; b: [64 lanes][KS] (fp32 slots);
    acc: [64][MT][16] (updated in place).  v_mfma_f32_32x32x16_bf16: lane l supplies A[l&31][8(l>>5)+e] and
    B[8(l>>5)+e][l&31], e < 8; the three pieces of each operand sum back to the fp32 value, so the emulation works
    on the reassembled values (the kernel's six-product expansion differs from that only at the 2^-23 level)."""
    KS8 = (KS + 7) // 8
    blk = block_weights(block, MT, KS8)                                                                # [mt][g][lane][e]
    bp = np.zeros((64, KS8 * 8))
    bp[:, :KS] = b[:, :KS]
    for mt in range(MT):
        D = np.zeros((32, 32))
        for g in range(KS8):
            for h in range(2):
                A = blk[mt, g, 32 * h:32 * h + 32, :]             # [i][e]
                B = bp[32 * h:32 * h + 32, 8 * g:8 * g + 8].T      # [e][j]
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


# ---- quad layout (csrc/mlp16.hpp): v_mfma_f32_16x16x32_bf16, lane = point j + 16 * quarter q ---------------------------
def gemm16(block, MT, KG, b, acc):
    """block: packed weights [MT][KG][3 pieces][64 lanes][8 bf16] as float32 words; b: [64 lanes][8*KG] fp32 k-values (lane
    (j, kq) supplies B[k = (g, kq, e)][j] = b[lane, 8g+e]); acc: [64][MT][4] updated in place (lane (j, q), register r =
    D[row 4q + r][col j] of each 16-row output tile).  The MFMA pairs element e of quarter kq of A with element e of
    quarter kq of B, whatever the hardware's internal k order."""
    blk = block_weights(block, MT, KG)                                                                 # [mt][g][lane][e]
    for mt in range(MT):
        D = np.zeros((16, 16))
        for g in range(KG):
            for kq in range(4):
                A = blk[mt, g, 16 * kq:16 * kq + 16, :]               # [i][e]
                B = b[16 * kq:16 * kq + 16, 8 * g:8 * g + 8].T        # [e][j]
                D += A @ B
        for lane in range(64):
            j, q = lane & 15, lane >> 4
            for r in range(4):
                acc[lane, mt, r] += D[4 * q + r, j]


def load_vec16(vec):
    """activation-layout vector [q*16 + s] -> [64 lanes][4 tiles][4]"""
    v = np.asarray(vec, dtype=np.float64)
    out = np.zeros((64, 4, 4))
    for lane in range(64):
        out[lane] = v[(lane >> 4) * 16:(lane >> 4) * 16 + 16].reshape(4, 4)
    return out


def quad_sum(v):
    """sum over the four quarter-lanes of a point"""
    s = v.reshape(4, 16).sum(0)
    return np.tile(s, 4)
