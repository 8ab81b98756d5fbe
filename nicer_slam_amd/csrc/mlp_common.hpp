// mlp_common.hpp -- register-resident batched MLP on the gfx950 matrix cores for the fused render-core kernels.
//
// Tiling (wave64, v_mfma_f32_32x32x16_bf16 with fp32 accumulation; fp32 operands enter as three exact bf16 pieces, see
// the GEMM primitive below):
//   * a wave owns a tile of 32 points; lane l works for point (l & 31), and the two half-waves h = l >> 5 split every
//     per-point job in two (half of the grid levels, half of the positional-encoding pairs, half of every activation
//     vector);
//   * a layer is D[out, point] += W[out, k] * act[k, point]: the WEIGHTS are the MFMA "A" operand (M = output
//     features, 32 per tile) and the ACTIVATIONS the "B" operand (N = points).  The MFMA result layout gives
//     lane (p, h), register r of output tile t the feature  f = 32 t + (r & 3) + 8 (r >> 2) + 4 h  of point p.
//     That register IS the B operand of k-slot  s = 16 t + r  of the next layer (k-order inside a GEMM is free as
//     long as A agrees), so activations chain in registers with no transposes and no LDS round trips;
//   * the weights are pre-packed on the device into that fragment order (fused/pack.py: slot groups of 8, three bf16
//     pieces, lane-minor 16-byte fragments).  Kernels either stream the fragments per wave from L2 (gemm_op: sampler,
//     colour forward) or stage each layer's block once per workgroup in LDS with asynchronous global->LDS copies
//     (gemm_staged / gemm_staged_part: SDF networks, colour backward).
//
// Packed per-feature vectors (biases, the sdf row) in activation layout:   idx = (t*2 + h)*16 + r.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nsa {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int HID = 64;        // hidden width of every MLP in the shipped configs
constexpr int HS = 32;         // k-steps of a hidden activation (64 features / 2 half-waves)
constexpr int SDF_IN_STEPS = 36;   // 71 inputs -> 36 slots per half-wave (one zero pad)
constexpr int COL_IN_STEPS = 65;   // 129 inputs -> 65 slots per half-wave (one zero pad)

// ---------------------------------------------------------------------------------------------------------------
// GEMM primitive.  acc[mt] (32 output features x 32 points, fp32) += sum_slots A(mt, slot) * b[slot].
//
// fp32-faithful products on the bf16 matrix cores.  Every fp32 operand is split exactly into
// three bf16 pieces (8 + 8 + 8 significand bits: x = hi + mid + lo), and the six cross products whose weight is
// >= 2^-16 are accumulated in fp32 by v_mfma_f32_32x32x16_bf16:
//     a*b ~= a_lo b_hi + a_hi b_lo + a_mid b_mid + a_mid b_hi + a_hi b_mid + a_hi b_hi          (error <= ~2^-23 |a b|)
// Each bf16 product is exact in fp32, so the result is fp32-class (measured against float64 it is no worse than an fp32
// fmaf chain).  Why: the fp32-input MFMA runs at the fp32 VECTOR rate and -- measured, SQ_VALU_MFMA_COEXEC_CYCLES = 0 and
// MFMA-busy + VALU-active == busy cycles -- does not overlap with VALU work at all, while one 32x32x16 bf16 MFMA does 8x
// the MACs in half the cycles: 6 of them replace 8 fp32 MFMAs (2.7x fewer matrix cycles) and run beside the VALU.
// Weights are split on the host (fused/pack.py); activations are split here with 2 ANDs + 2 SUBs + 1.5 PERMs per value.
// (The plain fp32-input MFMA variant, v_mfma_f32_32x32x2_f32, measured 1.145 vs 1.005 ms per iteration before it was
// removed; profiles/r01_pmc_per_kernel_v1.csv holds its counters.)
//
// Packed "A" block (weights), KS8 = ceil(KS/8) slot groups:   [mt][group][piece hi/mid/lo][lane][8 bf16]
//   lane l = (row i = l & 31, half h = l >> 5) holds W[row(mt,i)][slot(8g+e, h)], e = 0..7  -- the 8 k-values that the
//   MFMA takes from that lane; the activation fragment of lane (p, h) is its own slots 8g..8g+7.
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;

__host__ __device__ constexpr int a_block_floats(int mt, int ks) { return mt * ((ks + 7) / 8) * 3 * 64 * 4; }

struct BFrag {
    uint4 p[3];   // hi, mid, lo: 8 bf16 each
};

__device__ __forceinline__ bf16x8_t as_bf16x8(const uint4& v) {
    bf16x8_t r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}

// exact 3-way split of 8 fp32 values into packed bf16 fragments (truncation split: every piece is the top 16 bits)
__device__ __forceinline__ void split8(const float (&x)[8], BFrag& f) {
    unsigned u0[8], u1[8], u2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        u0[e] = __float_as_uint(x[e]);
        const float r1 = x[e] - __uint_as_float(u0[e] & 0xFFFF0000u);
        u1[e] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(u1[e] & 0xFFFF0000u);
        u2[e] = __float_as_uint(r2);
    }
#define NSA_PK(u, d) __builtin_amdgcn_perm(u[2 * d + 1], u[2 * d], 0x07060302u)
    f.p[0] = make_uint4(NSA_PK(u0, 0), NSA_PK(u0, 1), NSA_PK(u0, 2), NSA_PK(u0, 3));
    f.p[1] = make_uint4(NSA_PK(u1, 0), NSA_PK(u1, 1), NSA_PK(u1, 2), NSA_PK(u1, 3));
    f.p[2] = make_uint4(NSA_PK(u2, 0), NSA_PK(u2, 1), NSA_PK(u2, 2), NSA_PK(u2, 3));
#undef NSA_PK
}

// Operand precision of every GEMM in a translation unit: 3 = fp32-faithful split (default), 1 = plain bf16 operands
// (round-to-nearest-even; fp32 accumulate) -- the optional "bf16 MLP" mode, compiled as a second set of kernels
// (csrc/*_bf16.hip) and selected per network through nsa_grid_t.precision.  The packed weight blocks are shared: the
// bf16 mode reads only the round-to-nearest bf16 piece (form 3: the first slot of a fragment triple, form 2: the third; kSlot0).
// NSA_FORM: how the library's fp32 path enters the matrix cores (one choice per library build; fused/pack.py asks nsa_operand_form()):
//   3  three exact bf16 pieces per operand, six products per block (rounds 1-6a)
//   2  two fp16 pieces per operand (round to nearest, twice: x s = h0 + h1 to 2^-23), FOUR products per block, the weights scaled by
//      2^9 in the pack and every point's B vector by its own power of two (point_scale below): each operand is held to 2^-23, the
//      four products are exact, so a product is within 2^-22 |a b| in the worst case (three-piece form: 2^-23; an fp32 multiply: 2^-24)
//      with a third fewer matrix instructions and a third fewer accumulator roundings -- measured against float64 its dot products
//      are as close as the three-piece form's (emulation: closer; MI355X: 1.34e-7 against 1.22e-7 rms on sdf values of magnitude 1) and
//      closer than an fp32 multiply-add chain's (1.75e-7) -- tests/test_operand_form_cpu.py, test_operand_form_gpu.py, DESIGN 4.4
//      -DNSA_FORM2_PRODUCTS=3 (a build-time option, NOT the default) leaves the h1 h1 product out: 2^-22 |a b| more per product in the worst
//      case, a quarter fewer matrix instructions (-4.9 % per iteration); on the MI355X its sdf values are as close to float64 as the
//      four-product form's (1.35e-7 against 1.34e-7 rms) and every parity test passes at unchanged tolerances (profiles/r06_ab_experiments.txt r7p)
#ifndef NSA_FORM
#define NSA_FORM 2
#endif
#ifndef NSA_FORM2_PRODUCTS
#define NSA_FORM2_PRODUCTS 4
#endif
static_assert(NSA_FORM2_PRODUCTS == 4 || NSA_FORM2_PRODUCTS == 3, "NSA_FORM2_PRODUCTS: 4 (default) or 3");
#ifndef NSA_PIECES
#define NSA_PIECES NSA_FORM
#endif
constexpr int kForm = NSA_FORM;
constexpr int kPieces = NSA_PIECES;
// A packed fragment triple holds [bf16 hi | bf16 mid | bf16 lo] (form 3) or [fp16 h0 | fp16 h1 | bf16 round-to-nearest] (form 2: the
// third slot serves the bf16-operand build).  kSlot0 = the first slot this translation unit's pieces come from.
constexpr int kSlot0 = (NSA_PIECES == 1 && NSA_FORM == 2) ? 2 : 0;
constexpr float kWScale = 512.0f;            // form 2: packed weights are w * 2^9 (|w| < 127.9; held to 2^-23 for |w| >= 2^-11, to 2^-34 absolute below)
// source fragment (of the three per slot group and tile) of the i-th fragment a stage keeps in LDS
__host__ __device__ constexpr int src_frag(int i);
// Pieces per fragment that a block-cooperative weight stage keeps in LDS.  The packed blocks in global memory always hold all three
// (one pack serves both precisions); the bf16-operand build multiplies with one piece only, so its stages copy every THIRD 1-KiB
// fragment and lay the copy out as [tile][group][lane] -- a third of the global->LDS traffic behind every staged GEMM, which in that
// build is six times shorter and would otherwise wait for its copy (round 5).  NSA_STAGE_ALL_PIECES: the round-4 behaviour (A/B runs).
#ifdef NSA_STAGE_ALL_PIECES
constexpr int kLdsPieces = 3;
#else
constexpr int kLdsPieces = NSA_PIECES;
#endif
__host__ __device__ constexpr int src_frag(int i) {
    return kLdsPieces == 3 ? i : (kLdsPieces == 2 ? 3 * (i >> 1) + (i & 1) : 3 * i + kSlot0);
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16x8_t round8_bf16(const float (&x)[8]) {     // v_cvt_pk_bf16_f32 x 4
    unsigned u[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const f32x2_t v = {x[2 * d], x[2 * d + 1]};
        const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
        __builtin_memcpy(&u[d], &b, 4);
    }
    bf16x8_t r;
    __builtin_memcpy(&r, u, 16);
    return r;
}

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;      // (uint4 is a class type: no address-space pointers to it)
using lds_u4 = __attribute__((address_space(3))) u32x4;

__device__ __forceinline__ bf16x8_t as_bf16x8(const u32x4& v) {
    bf16x8_t r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}

// ---- form 2: two fp16 pieces ------------------------------------------------------------------------------------------------------
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
template <class V>
__device__ __forceinline__ f16x8_t as_f16x8(const V& v) {
    f16x8_t r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}
// t = h0 + h1 (+ at most 2^-23 |t|): h0 = fp16(t), h1 = fp16(t - h0), both round-to-nearest (v_cvt_pk_f16_f32); the subtraction is
// exact in fp32.  t is already scaled (point_scale): |t| < 2^14, so nothing overflows, and an h1 below fp16's normal range costs at
// most 2^-25 absolute, i.e. 2^-38 of the point's largest operand.  (Scalar subtractions on purpose: a float2 expression would
// compile to v_pk_add_f32, which build.py's ISA check refuses)
__device__ __forceinline__ void split8_h2(const float (&t)[8], f16x8_t& h0, f16x8_t& h1) {
    unsigned u0[4], u1[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const f32x2_t v = {t[2 * d], t[2 * d + 1]};
        const f16x2_t p = __builtin_convertvector(v, f16x2_t);
        const float r0 = t[2 * d] - (float)p[0];
        const float r1 = t[2 * d + 1] - (float)p[1];
        const f32x2_t rv = {r0, r1};
        const f16x2_t q = __builtin_convertvector(rv, f16x2_t);
        __builtin_memcpy(&u0[d], &p, 4);
        __builtin_memcpy(&u1[d], &q, 4);
    }
    __builtin_memcpy(&h0, u0, 16);
    __builtin_memcpy(&h1, u1, 16);
}

// Per-point power-of-two scaling of a GEMM's B operand (form 2).  m = the largest |b| of the point (over all of its lanes; the caller
// has combined the lanes' maxima): the operands are multiplied by s = 2^(13 - E), E = exponent of m clamped to +-80, so that the
// largest lies in [2^13, 2^14); the accumulators (which hold the bias or an earlier partial sum) are multiplied by s 2^9 before the
// products are added and by its inverse afterwards -- exact (powers of two), and the fp32 additions in between round as they would
// unscaled.  An all-zero vector scales by 2^93 (harmless).
struct PointScale {
    float s;        // operands x s
    int kpre;       // accumulators x 2^kpre before, x 2^-kpre after (v_ldexp_f32: a multiplication of the accumulator VECTORS by a
};                  // float would compile to v_pk_mul_f32, which build.py's ISA check refuses)
__device__ __forceinline__ PointScale point_scale_of(float m) {
    unsigned e = __float_as_uint(m) >> 23;               // biased exponent (m >= 0)
    e = e < 47u ? 47u : (e > 207u ? 207u : e);          // (|E| <= 80: an accumulator of magnitude 2^25 still survives the 2^(149 - e) below)
#ifdef NSA_X_NO_POINT_SCALE      // timing-only ablation (WRONG numbers; tagged builds): no per-point maximum, a constant scale (r6w)
    e = 127u;
#endif
    PointScale r;
    r.s = __uint_as_float((267u - e) << 23);             // 2^(140 - e)
    r.kpre = 149 - (int)e;                               // s * 2^9
    return r;
}
template <int N>
__device__ __forceinline__ float abs_max(const float (&b)[N]) {
    float m[3] = {0.0f, 0.0f, 0.0f};                    // three independent chains of v_max3_f32 (|.| modifiers are free)
#pragma unroll
    for (int k = 0; k + 1 < N; k += 2) m[(k >> 1) % 3] = fmaxf(fmaxf(m[(k >> 1) % 3], fabsf(b[k])), fabsf(b[k + 1]));
    if (N & 1) m[0] = fmaxf(m[0], fabsf(b[N - 1]));
    return fmaxf(fmaxf(m[0], m[1]), m[2]);
}
// A bound on this lane's share of the NEXT GEMM's operands that is known before they are: Softplus(beta = 100) of an accumulator is at
// most |a| + ln 2 / 100, a ReLU at most |a|.  A GEMM that is handed such a bound (`hint`) takes the point's scale from it instead of
// from the operands themselves, so the operand conversion of its first slot group does not have to wait for the activation of the last
// value (the scale is a power of two: a bound that is loose by 2^k only moves the 2^-38 absolute floor of the split by 2^k).
template <int MT>
__device__ __forceinline__ float acc_abs_max(const f32x16 (&acc)[MT]) {
    float m[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 16 * MT; i += 2) m[(i >> 1) % 3] = fmaxf(fmaxf(m[(i >> 1) % 3], fabsf(acc[i >> 4][i & 15])), fabsf(acc[i >> 4][(i + 1) & 15]));
    return fmaxf(fmaxf(m[0], m[1]), m[2]);
}
constexpr float kSoftplusSlack = 0.00694f;      // > ln 2 / 100

// max of a non-negative float over the lane pairs (l, l ^ 32) / quads (l, l ^ 16, l ^ 32, l ^ 48): v_permlane32_swap / v_permlane16_swap
// (gfx950; vector ALU, no LDS round trip) and integer maxima of the bit patterns (non-negative floats order like unsigned integers)
__device__ __forceinline__ unsigned umax_xor32(unsigned m) {
    const auto r = __builtin_amdgcn_permlane32_swap(m, m, false, false);
    return r[0] > r[1] ? r[0] : r[1];
}
__device__ __forceinline__ unsigned umax_xor16(unsigned m) {
    const auto r = __builtin_amdgcn_permlane16_swap(m, m, false, false);
    return r[0] > r[1] ? r[0] : r[1];
}
// 32-point tiling: a point's operands live in lanes p and p + 32
template <int N>
__device__ __forceinline__ PointScale point_scale32(const float (&b)[N], const float* hint = nullptr) {
    const float m = hint ? *hint : abs_max<N>(b);
    return point_scale_of(__uint_as_float(umax_xor32(__float_as_uint(m))));
}
template <int MT, bool POST>      // POST: the scaling behind the products (false: the one in front of them)
__device__ __forceinline__ void scale_acc(f32x16 (&acc)[MT], int k) {
#ifdef NSA_X_NO_ACC_SCALE        // timing-only ablations (WRONG numbers; tagged builds): what the accumulator scaling of form 2 costs (r6w),
    return;                      // and its two halves (r7i)
#endif
#ifdef NSA_X_NO_PRE
    if (!POST) return;
#endif
#ifdef NSA_X_NO_POST
    if (POST) return;
#endif
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = __builtin_ldexpf(acc[mt][r], k);
}

// acc[mt] += A(mt, group) * x for one slot group: the six cross products of the 3-way split, the four of the two-piece fp16 form (x
// already scaled), or one bf16 product.
template <int MT, class AV>
__device__ __forceinline__ void mma_group(const AV (&a)[MT][3], const float (&x)[8], f32x16 (&acc)[MT]) {
    if constexpr (kPieces == 3) {
        BFrag bf;
        split8(x, bf);
        const bf16x8_t bh = as_bf16x8(bf.p[0]), bm = as_bf16x8(bf.p[1]), bl = as_bf16x8(bf.p[2]);
        // smallest terms first; the MT accumulators alternate so no MFMA waits on its predecessor
#define NSA_MM(AP, BV)                                                                                   \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                \
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[mt][AP]), BV, acc[mt], 0, 0, 0);
        NSA_MM(2, bh) NSA_MM(0, bl) NSA_MM(1, bm) NSA_MM(1, bh) NSA_MM(0, bm) NSA_MM(0, bh)
#undef NSA_MM
    } else if constexpr (kPieces == 2) {
        f16x8_t b0, b1;
        split8_h2(x, b0, b1);
#define NSA_MMH(AP, BV)                                                                                  \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                \
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(a[mt][AP]), BV, acc[mt], 0, 0, 0);
#if NSA_FORM2_PRODUCTS == 4      // (-DNSA_FORM2_PRODUCTS=3: without the h1 h1 product, see the NSA_FORM comment)
        NSA_MMH(1, b1)
#endif
        NSA_MMH(0, b1) NSA_MMH(1, b0) NSA_MMH(0, b0)
#undef NSA_MMH
    } else {
        const bf16x8_t b = round8_bf16(x);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[mt][0]), b, acc[mt], 0, 0, 0);
    }
}

template <int MT>
struct AFrag {
    uint4 g[MT][3];   // the next slot group's weight pieces
};

template <int KS, int MT>
__device__ __forceinline__ void gemm_preload(const float* __restrict__ wp, int lane, AFrag<MT>& f) {
    const uint4* __restrict__ w4 = reinterpret_cast<const uint4*>(wp) + lane;
    constexpr int KS8 = (KS + 7) / 8;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int pc = 0; pc < kPieces; ++pc) f.g[mt][pc] = w4[((mt * KS8 + 0) * 3 + kSlot0 + pc) * 64];
}

template <int KS, int MT>
__device__ __forceinline__ void gemm_run(const float* __restrict__ wp, int lane, AFrag<MT>& f, const float (&b)[KS],
                                         f32x16 (&acc)[MT], const float* hint = nullptr) {
    constexpr int KS8 = (KS + 7) / 8;
    const uint4* __restrict__ w4 = reinterpret_cast<const uint4*>(wp) + lane;
    PointScale ps{1.0f, 0};
    if constexpr (kPieces == 2) {
        ps = point_scale32<KS>(b, hint);
        scale_acc<MT, false>(acc, ps.kpre);
    }
#pragma unroll
    for (int g = 0; g < KS8; ++g) {
        uint4 a[MT][3];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int pc = 0; pc < kPieces; ++pc) {
                a[mt][pc] = f.g[mt][pc];
                if (g + 1 < KS8) f.g[mt][pc] = w4[((mt * KS8 + g + 1) * 3 + kSlot0 + pc) * 64];
            }
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (8 * g + e < KS) ? (kPieces == 2 ? b[8 * g + e] * ps.s : b[8 * g + e]) : 0.0f;
        mma_group<MT>(a, x, acc);
    }
    if constexpr (kPieces == 2) scale_acc<MT, true>(acc, -ps.kpre);
    __builtin_amdgcn_sched_barrier(0);   // keep later layers' loads from being hoisted above this GEMM
}

// ---- block-cooperative weight staging -------------------------------------------------------------------------
// Measured (profiles/r01_*, an ablation build of round 1 in which every fragment load hit the same 3 KB): with every wave streaming its own copy of the packed weights from
// L2, the fine SDF backward spends 59 % of its wave cycles in s_waitcnt and 127 of its 280 us disappear when the
// fragment loads hit L1 -- at one wave per SIMD there is nothing to hide the L2 latency behind, and the four waves of a
// block fetch the same bytes four times.  The staged GEMM fetches each layer's packed block ONCE per workgroup with
// asynchronous global->LDS copies (global_load_lds_dwordx4: no VGPR staging; 1 KiB per wave instruction, which is
// exactly one fragment), double-buffered one GEMM ahead, and the MFMA A operands are read from LDS (ds_read_b128).
// One workgroup barrier per GEMM.  All four waves of a block must execute the same GEMM sequence (no early exits).
constexpr int kStageFloats = 9216;      // largest packed block: A[3 tiles][32 slots] = 36 KiB



__device__ __forceinline__ void stage_issue(const float* __restrict__ g, int nfloats, float* lds_dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = nfloats / 256 / 3 * kLdsPieces;      // 1-KiB fragments copied (a packed A block is a multiple of 3 fragments)
#pragma unroll
    for (int c = 0; c < (chunks + 3) / 4; ++c) {
        const int ch = 4 * c + wave;
        if (ch < chunks)                                    // (src_frag: the pieces this build multiplies with, of the three per fragment triple)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + src_frag(ch) * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(lds_dst + ch * 256), 16, 0, 0);
    }
}

#ifdef NSA_X_TS       // profiling build only (tools/ts_profile.py): where a wave's cycles go, accumulated per wave in LDS
static __shared__ unsigned long long nsa_ts_lds[16][16];
__device__ __forceinline__ void ts_add(int slot, unsigned long long dt) {
    if ((threadIdx.x & 63) == 0) nsa_ts_lds[threadIdx.x >> 6][slot] += dt;
}
__device__ __forceinline__ unsigned long long ts_now() { return __builtin_readcyclecounter(); }
#endif
__device__ __forceinline__ void stage_wait() {
#ifdef NSA_X_TS
    const unsigned long long t0 = ts_now();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const unsigned long long t1 = ts_now();
    __syncthreads();
    const unsigned long long t2 = ts_now();
    ts_add(0, t1 - t0);
    ts_add(1, t2 - t1);
    ts_add(2, 1);
#else
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): this wave's async copies (and older global loads) have landed
    __syncthreads();                         // ... and so have everyone else's; all waves are done with the other buffer
#endif
}

template <int KS, int MT>
__device__ __forceinline__ void gemm_lds(const float* lds_block, int lane, const float (&b)[KS], f32x16 (&acc)[MT],
                                         const float* hint = nullptr) {
    constexpr int KS8 = (KS + 7) / 8;
    const lds_u4* w4 = (const lds_u4*)lds_block + lane;
    PointScale ps{1.0f, 0};
    if constexpr (kPieces == 2) {
        ps = point_scale32<KS>(b, hint);
        scale_acc<MT, false>(acc, ps.kpre);
    }
    u32x4 nxt[MT][3];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int pc = 0; pc < kPieces; ++pc) nxt[mt][pc] = w4[((mt * KS8 + 0) * kLdsPieces + pc) * 64];
#pragma unroll
    for (int g = 0; g < KS8; ++g) {
        u32x4 a[MT][3];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int pc = 0; pc < kPieces; ++pc) {
                a[mt][pc] = nxt[mt][pc];
                if (g + 1 < KS8) nxt[mt][pc] = w4[((mt * KS8 + g + 1) * kLdsPieces + pc) * 64];
            }
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (8 * g + e < KS) ? (kPieces == 2 ? b[8 * g + e] * ps.s : b[8 * g + e]) : 0.0f;
        mma_group<MT>(a, x, acc);
    }
    if constexpr (kPieces == 2) scale_acc<MT, true>(acc, -ps.kpre);
    __builtin_amdgcn_sched_barrier(0);
}

// GEMM number `opi` of the kernel's sequence Seq (Seq::n ops; Seq::off(i) / Seq::size(i) = packed block of op i, in
// floats from `wp`): wait for its block, start fetching the next one into the other buffer, multiply from LDS.
template <class Seq, int KS, int MT>
__device__ __forceinline__ void gemm_staged(float* stage, const float* __restrict__ wp, int opi, int lane,
                                            const float (&b)[KS], f32x16 (&acc)[MT], const float* hint = nullptr) {
    stage_wait();
    if (opi + 1 < Seq::n) stage_issue(wp + Seq::off(opi + 1), Seq::size(opi + 1), stage + ((opi + 1) & 1) * kStageFloats);
    gemm_lds<KS, MT>(stage + (opi & 1) * kStageFloats, lane, b, acc, hint);
}

template <class Seq>
__device__ __forceinline__ void stage_begin(float* stage, const float* __restrict__ wp) {
    stage_issue(wp + Seq::off(0), Seq::size(0), stage);
}

// ---- staged GEMMs split along k (packed blocks larger than a stage buffer) -------------------------------------------
// A packed block is [mt][group][piece][lane]; the part "groups G0 .. G0+NG-1 of every tile" is MT strided segments in
// global memory and is laid out in LDS as [mt][NG][piece][lane].  The parts of one GEMM accumulate into the same acc,
// and each slot group's B operand is split once.
struct StageOp {
    int off;    // block start, floats from wp
    int mt;     // tiles
    int ks8;    // slot groups of the whole block
    int g0, ng; // this part
};

__device__ __forceinline__ void stage_issue_op(const float* __restrict__ wp, const StageOp o, float* lds_dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_tile = o.ng * kLdsPieces;        // 1 KiB chunks per tile in this part
    const int chunks = o.mt * per_tile;
#pragma unroll
    for (int c = 0; c < (chunks + 3) / 4; ++c) {
        const int ch = 4 * c + wave;
        if (ch < chunks) {
            const int mt = ch / per_tile, rem = ch - mt * per_tile;
            const float* src = wp + o.off + ((mt * o.ks8 + o.g0) * 3 + src_frag(rem)) * 256 + lane * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds_dst + ch * 256), 16, 0, 0);
        }
    }
}

// PRE / POST (form 2): the first part of a GEMM scales the accumulators up, the last one back down; parts in between leave them in
// the scaled domain (every part derives the same scale from the same b / hint).
template <int KS, int MT, int G0, int NG, bool PRE = true, bool POST = true>
__device__ __forceinline__ void gemm_lds_part(const float* lds_block, int lane, const float (&b)[KS], f32x16 (&acc)[MT],
                                              const float* hint = nullptr) {
    const lds_u4* w4 = (const lds_u4*)lds_block + lane;
    PointScale ps{1.0f, 0};
    if constexpr (kPieces == 2) {           // (every part of a GEMM sees the same b, hence the same scale)
        ps = point_scale32<KS>(b, hint);
        if constexpr (PRE) scale_acc<MT, false>(acc, ps.kpre);
    }
    u32x4 nxt[MT][3];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int pc = 0; pc < kPieces; ++pc) nxt[mt][pc] = w4[((mt * NG + 0) * kLdsPieces + pc) * 64];
#pragma unroll
    for (int gl = 0; gl < NG; ++gl) {
        const int g = G0 + gl;
        u32x4 a[MT][3];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int pc = 0; pc < kPieces; ++pc) {
                a[mt][pc] = nxt[mt][pc];
                if (gl + 1 < NG) nxt[mt][pc] = w4[((mt * NG + gl + 1) * kLdsPieces + pc) * 64];
            }
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (8 * g + e < KS) ? (kPieces == 2 ? b[8 * g + e] * ps.s : b[8 * g + e]) : 0.0f;
        mma_group<MT>(a, x, acc);
    }
    if constexpr (kPieces == 2 && POST) scale_acc<MT, true>(acc, -ps.kpre);
    __builtin_amdgcn_sched_barrier(0);
}

// Part `opi` of the kernel's sequence Seq (Seq::n parts, Seq::op(i) their descriptors); buffers of BUF floats.
template <class Seq, int BUF, int KS, int MT, int G0, int NG, bool PRE = true, bool POST = true>
__device__ __forceinline__ void gemm_staged_part(float* stage, const float* __restrict__ wp, int opi, int lane,
                                                 const float (&b)[KS], f32x16 (&acc)[MT], const float* hint = nullptr) {
    stage_wait();
    if (opi + 1 < Seq::n) stage_issue_op(wp, Seq::op(opi + 1), stage + ((opi + 1) & 1) * BUF);
    gemm_lds_part<KS, MT, G0, NG, PRE, POST>(stage + (opi & 1) * BUF, lane, b, acc, hint);
}




template <int KS, int MT>
__device__ __forceinline__ void gemm_op(const float* __restrict__ wp, int lane, const float (&b)[KS], f32x16 (&acc)[MT],
                                        const float* hint = nullptr) {
    AFrag<MT> f;
    gemm_preload<KS, MT>(wp, lane, f);
    gemm_run<KS, MT>(wp, lane, f, b, acc, hint);
}

// The same for T point tiles per wave: one weight fragment stream feeds T independent accumulator sets, so the fragment traffic
// (3 KiB per slot group and output tile, the L1's whole 64 B/clk when the wave is MFMA-bound) is paid once per T tiles, and the
// operand split of tile t + 1 issues while the matrix cores work on tile t.
template <int KS, int MT, int T>
__device__ __forceinline__ void gemm_op_tiles(const float* __restrict__ wp, int lane, const float (&b)[T][KS], f32x16 (&acc)[T][MT],
                                              const float* hint = nullptr) {       // hint[t]: see acc_abs_max
    constexpr int KS8 = (KS + 7) / 8;
    const uint4* __restrict__ w4 = reinterpret_cast<const uint4*>(wp) + lane;
    AFrag<MT> f;
    gemm_preload<KS, MT>(wp, lane, f);
    PointScale ps[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        ps[t] = PointScale{1.0f, 0};
        if constexpr (kPieces == 2) {
            ps[t] = point_scale32<KS>(b[t], hint ? hint + t : nullptr);
            scale_acc<MT, false>(acc[t], ps[t].kpre);
        }
    }
#pragma unroll
    for (int g = 0; g < KS8; ++g) {
        uint4 a[MT][3];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int pc = 0; pc < kPieces; ++pc) {
                a[mt][pc] = f.g[mt][pc];
                if (g + 1 < KS8) f.g[mt][pc] = w4[((mt * KS8 + g + 1) * 3 + kSlot0 + pc) * 64];
            }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (8 * g + e < KS) ? (kPieces == 2 ? b[t][8 * g + e] * ps[t].s : b[t][8 * g + e]) : 0.0f;
            mma_group<MT>(a, x, acc[t]);
        }
    }
    if constexpr (kPieces == 2) {
#pragma unroll
        for (int t = 0; t < T; ++t) scale_acc<MT, true>(acc[t], -ps[t].kpre);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// Load a packed per-feature vector (bias) for MT tiles into accumulator layout.
template <int MT>
__device__ __forceinline__ void load_vec(const float* __restrict__ vp, int h, f32x16 (&acc)[MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const float4* p = reinterpret_cast<const float4*>(vp + (mt * 2 + h) * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 v = p[i];
            acc[mt][4 * i + 0] = v.x; acc[mt][4 * i + 1] = v.y; acc[mt][4 * i + 2] = v.z; acc[mt][4 * i + 3] = v.w;
        }
    }
}

// Two waves share a SIMD's matrix pipe and VALU issue port.  Running the same program from the same start they stay
// in lock-step (both in a VALU phase, then both in an MFMA phase) and the two pipes never overlap -- measured:
// MFMA-busy + VALU-active cycles == wave cycles.  Giving the co-resident waves different static priorities makes the
// favoured wave run ahead; the other fills the slots it leaves and the phases become complementary (matrix beside
// VALU).  The wave slot id (HW_REG_HW_ID[3:0]) differs between the waves of one SIMD.
__device__ __forceinline__ void desync_simd_partners() {
    const unsigned slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);   // hwreg(HW_REG_HW_ID, 0, 4) = WAVE_ID
    if (slot & 1u) __builtin_amdgcn_s_setprio(1);
    else           __builtin_amdgcn_s_setprio(0);
}

__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32); }

// Softplus(beta = 100) with PyTorch's threshold 20 (base_networks.py:153), its derivative sigmoid(100 a) and the
// second derivative 100 s (1 - s) (zero in the linear region, like torch's double backward).
// Softplus(beta = 100) with PyTorch's threshold 20 (base_networks.py:153), its derivative sigmoid(100 a) and the second
// derivative 100 s (1 - s) (zero in the linear region, like torch's double backward).  exp / log go through the raw
// base-2 v_exp_f32 / v_log_f32 with the constants folded in: arguments stay in [.., 2^29] resp. [1, 2^29], so the library
// wrappers' denormal fix-ups (v_ldexp + compares) would be dead weight.
constexpr float SP_K = 100.0f * 1.4426950408889634f;       // beta * log2(e)
constexpr float SP_LIN = 20.0f * 1.4426950408889634f;      // threshold in the same units
constexpr float SP_OUT = 0.01f * 0.6931471805599453f;      // ln(2) / beta

// max(a, 0).  fmaxf compiles to TWO v_max_f32 (a canonicalising v_max a, a in front: IEEE-mode signalling-NaN quieting the
// compiler cannot prove unnecessary for an MFMA result).  One-instruction forms were measured (profiles/r03_ab_experiments.txt):
// inline asm -- WRONG results, the hazard recogniser cannot see an asm operand and a VALU read of a fresh MFMA result without
// its wait states returns stale data; integer max on the bit pattern -- correct, 128 fewer VALU per 32 sampler points, no time
// gained there and +9 us in the colour forward.  So: fmaxf.
__device__ __forceinline__ float relu_f(float a) { return fmaxf(a, 0.0f); }

// value only (sampler): the overflow-free form max(a,0) + ln(1 + e^{-|beta a|})/beta -- no compare/select, and equal to
// torch's thresholded softplus to the last ulp (for beta a > 20 the log term is < 2e-9 relative and rounds away).
// NSA_X_SOFTPLUS_FREE (tagged experiment builds only, build.py rejects NSA_X_* in the product): the activation and its derivatives
// replaced by ReLU's -- WRONG numbers, timing only: the upper bound of what any cheaper (packed, approximated, table-driven) Softplus
// could give a kernel (profiles/r06_ab_experiments.txt r6f).
__device__ __forceinline__ float softplus100(float a) {
#ifdef NSA_X_SOFTPLUS_FREE
    return relu_f(a);
#endif
    const float t = SP_K * a;
    return fmaf(SP_OUT, __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(t))), relu_f(a));
}
__device__ __forceinline__ float softplus100_d1(float a) {
#ifdef NSA_X_SOFTPLUS_FREE
    return a > 0.0f ? 1.0f : 0.0f;
#endif
    const float t = SP_K * a;
    const float e = __builtin_amdgcn_exp2f(t);
    return t > SP_LIN ? 1.0f : e * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ void softplus100_all(float a, float& y, float& d1, float& d2) {
#ifdef NSA_X_SOFTPLUS_FREE
    y = relu_f(a); d1 = a > 0.0f ? 1.0f : 0.0f; d2 = 0.0f;
    return;
#endif
    const float t = SP_K * a;
    const float e = __builtin_amdgcn_exp2f(t);
    const bool lin = t > SP_LIN;
    const float s = e * __builtin_amdgcn_rcpf(e + 1.0f);
    y = lin ? a : SP_OUT * __builtin_amdgcn_logf(1.0f + e);
    d1 = lin ? 1.0f : s;
    d2 = lin ? 0.0f : 100.0f * s * (1.0f - s);
}

// sin and cos of a (|a| <~ 1e3) sharing one Cody-Waite reduction to [-pi/4, pi/4]; ~1 ulp-class polynomials.
__device__ __forceinline__ void sincos_f(float a, float& s, float& c) {
#ifdef NSA_X_PE_FREE      // (tagged experiment builds only; WRONG numbers, timing only: the bound of a cheaper positional encoding)
    s = a; c = 1.0f - a;
    return;
#endif
    const float n = rintf(a * 0.63661977236758134f);          // a / (pi/2)
    float r = fmaf(n, -1.5707963705062866f, a);                 // pi/2 = c1 + c2 + c3 (float32 parts)
    r = fmaf(n, 4.371138828673793e-08f, r);
    r = fmaf(n, 1.7763568394002505e-15f, r);
    const float r2 = r * r;
    float ps = fmaf(r2, 2.7557314297e-6f, -1.9841270114e-4f);   // sin r = r + r^3 * P(r^2)
    ps = fmaf(ps, r2, 8.3333337680e-3f);
    ps = fmaf(ps, r2, -1.6666667163e-1f);
    const float sr = fmaf(ps * r2, r, r);
    float pc = fmaf(r2, 2.4433157e-5f, -1.3887316255e-3f);      // cos r = 1 - r^2/2 + r^4 * Q(r^2)
    pc = fmaf(pc, r2, 4.1666645683e-2f);
    const float cr = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
    const int q = (int)n & 3;
    const float s0 = (q & 1) ? cr : sr;
    const float c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

}  // namespace nsa
