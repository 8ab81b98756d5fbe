// mlp16.hpp -- the "quad" tiling of the SDF-network kernels: v_mfma_f32_16x16x32_bf16, a wave = 16 points, FOUR lanes per
// point (lane = point j + 16 * quarter q).
//
// Why (measured on MI355X, tools/micro/mfma_issue.hip and the ablation builds of tools/build_ablations.sh, round 2):
//   * one wave per SIMD issues a VALU instruction every ~5.2 cycles, two or more waves every ~2.5: the 32-point tiling
//     (mlp_common.hpp; a lane PAIR per point) needs 256 .. 500 registers per lane in the fine-network kernels -- one or
//     two waves per SIMD, spills in the forward -- and those kernels ran at 36 .. 47 % of their issue-slot bound;
//   * per matrix instruction the hardware hides ~5 VALU instructions (32x32x16) resp. ~2 (16x16x32) and charges the rest
//     at the VALU rate, the same per MAC for both shapes -- the smaller shape costs nothing;
//   * four lanes per point halve every per-point array (activations, softplus derivatives, first-layer slots), which is
//     what buys the second / third wave per SIMD.
//
// Layout.  Result tile of one MFMA: D[16 out-features x 16 points], lane (j, q) register r = D[4q + r][j].  A 64-feature
// activation is therefore 16 floats per lane, index s = 4 t + r (t = output tile):
//     feature(s, q) = 16 (s >> 2) + 4 q + (s & 3)
// and k-group g (32 k-values, 8 per lane) of the next layer takes act[8g .. 8g+7] of every lane as its B operand: activations
// chain in registers exactly as in the 32-point tiling.  Packed A blocks (fused/pack.py::a_block16):
//     [out tile mt][k-group g][piece hi/mid/lo][lane][8 bf16]      -- 1 KiB per (mt, g, piece), lane (i, kq) holds
//     W[16 mt + i][k = (g, kq, e)], e = 0..7.
// Per-feature vectors (biases, the sdf row) in activation layout: idx = q * 16 + s.
#pragma once
#include "mlp_common.hpp"

namespace nsa {

using f32x4v = __attribute__((ext_vector_type(4))) float;

constexpr int QHS = 16;          // activation floats per lane (64 features / 4 quarter-lanes)
constexpr int QIN = 24;          // first-layer slots per lane: 3 k-groups
constexpr int QIN_G = 3;

__host__ __device__ constexpr int a16_floats(int mt, int kg) { return mt * kg * 3 * 64 * 4; }

// sum over the four quarter-lanes of a point (lanes j, j+16, j+32, j+48)
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}

// acc[t] += A(t, group) * x for TC output tiles of one k-group: the six cross products of the 3-way split (or one bf16 product);
// `bh/bm/bl` are the split pieces of the B operand (split once per k-group, shared by all output tiles)
template <int TC>
__device__ __forceinline__ void mma16_tiles(const u32x4 (&a)[TC][3], const bf16x8_t bh, const bf16x8_t bm, const bf16x8_t bl,
                                            f32x4v* acc) {
    if constexpr (kPieces == 3) {
#ifdef NSA_ABL_NOMFMA
        _Pragma("unroll") for (int t = 0; t < TC; ++t) acc[t][0] += __uint_as_float((a[t][0][0] ^ a[t][1][1] ^ a[t][2][2]) & 0x3F800000u);
        return;
#endif
#define NSA_MM16(AP, BV)                                                                               \
        _Pragma("unroll") for (int t = 0; t < TC; ++t)                                                 \
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a[t][AP]), BV, acc[t], 0, 0, 0);
        NSA_MM16(2, bh) NSA_MM16(0, bl) NSA_MM16(1, bm) NSA_MM16(1, bh) NSA_MM16(0, bm) NSA_MM16(0, bh)
#undef NSA_MM16
    } else {
#pragma unroll
        for (int t = 0; t < TC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a[t][0]), bh, acc[t], 0, 0, 0);
    }
}

// acc[MT] += A[MT x KG] * b, A block in LDS ([mt][g][piece][lane]), b = this lane's 8*KG k-values.
// Output tiles are processed in chunks of at most 4 (4 independent accumulators per product step: no MFMA waits on its
// predecessor; 12 fragment registers live per chunk instead of 3*MT).
template <int KG, int MT>
__device__ __forceinline__ void gemm16_lds(const float* lds_block, int lane, const float (&b)[8 * KG], f32x4v (&acc)[MT]) {
    const lds_u4* w4 = (const lds_u4*)lds_block + lane;
    constexpr int TC = MT <= 4 ? MT : (MT % 3 == 0 ? 3 : 4);
    static_assert(MT % TC == 0, "tile chunking");
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = b[8 * g + e];
        bf16x8_t bh, bm, bl;
        if constexpr (kPieces == 3) {
            BFrag bf;
            split8(x, bf);
            bh = as_bf16x8(bf.p[0]); bm = as_bf16x8(bf.p[1]); bl = as_bf16x8(bf.p[2]);
        } else {
            bh = round8_bf16(x); bm = bh; bl = bh;
        }
#pragma unroll
        for (int c = 0; c < MT / TC; ++c) {
            u32x4 a[TC][3];
#pragma unroll
            for (int t = 0; t < TC; ++t)
#pragma unroll
                for (int pc = 0; pc < kPieces; ++pc) a[t][pc] = w4[(((c * TC + t) * KG + g) * 3 + pc) * 64];
            mma16_tiles<TC>(a, bh, bm, bl, &acc[c * TC]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // keep later layers' LDS reads from being hoisted above this GEMM
}

// the same from global memory (per-wave streaming of the packed block; sampler fallback)
template <int KG, int MT>
__device__ __forceinline__ void gemm16_glb(const float* __restrict__ wp, int lane, const float (&b)[8 * KG], f32x4v (&acc)[MT]) {
    const uint4* __restrict__ w4 = reinterpret_cast<const uint4*>(wp) + lane;
    constexpr int TC = MT <= 4 ? MT : (MT % 3 == 0 ? 3 : 4);
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = b[8 * g + e];
        bf16x8_t bh, bm, bl;
        if constexpr (kPieces == 3) {
            BFrag bf;
            split8(x, bf);
            bh = as_bf16x8(bf.p[0]); bm = as_bf16x8(bf.p[1]); bl = as_bf16x8(bf.p[2]);
        } else {
            bh = round8_bf16(x); bm = bh; bl = bh;
        }
#pragma unroll
        for (int c = 0; c < MT / TC; ++c) {
            u32x4 a[TC][3];
#pragma unroll
            for (int t = 0; t < TC; ++t)
#pragma unroll
                for (int pc = 0; pc < kPieces; ++pc) {
                    const uint4 v = w4[(((c * TC + t) * KG + g) * 3 + pc) * 64];
                    a[t][pc] = u32x4{v.x, v.y, v.z, v.w};
                }
            mma16_tiles<TC>(a, bh, bm, bl, &acc[c * TC]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// ---- block-cooperative weight staging for NW waves per workgroup (see mlp_common.hpp for the 4-wave original) -------------
template <int NW>
__device__ __forceinline__ void stage_issue_n(const float* __restrict__ g, int nfloats, float* lds_dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = nfloats / 256;
    for (int ch = wave; ch < chunks; ch += NW)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + ch * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(lds_dst + ch * 256), 16, 0, 0);
}

// GEMM number `opi` of the kernel's sequence Seq: wait for its block, start fetching the next one into the other buffer,
// multiply from LDS.  All NW waves of the block must execute the same sequence.  (A ring of three buffers, fetching two GEMMs
// ahead behind a counted vmcnt, measured the same: the copies are not what the waves wait for -- profiles/r02_ab_experiments.txt.)
template <class Seq, int NW>
__device__ __forceinline__ void stage16_begin(float* stage, const float* __restrict__ wp) {
    stage_issue_n<NW>(wp + Seq::off(0), Seq::size(0), stage);
}

template <class Seq, int NW, int KG, int MT>
__device__ __forceinline__ void gemm16_staged(float* stage, const float* __restrict__ wp, int opi, int lane,
                                              const float (&b)[8 * KG], f32x4v (&acc)[MT]) {
    stage_wait();
    if (opi + 1 < Seq::n) stage_issue_n<NW>(wp + Seq::off(opi + 1), Seq::size(opi + 1), stage + ((opi + 1) & 1) * kStageFloats);
    gemm16_lds<KG, MT>(stage + (opi & 1) * kStageFloats, lane, b, acc);
}

// packed per-feature vector (activation layout [q*16 + s]) -> this lane's 16 values as 4 tiles x 4
__device__ __forceinline__ void load_vec16(const float* __restrict__ vp, int q, f32x4v (&acc)[4]) {
    const float4* p = reinterpret_cast<const float4*>(vp + q * 16);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float4 v = p[t];
        acc[t] = f32x4v{v.x, v.y, v.z, v.w};
    }
}

__device__ __forceinline__ void zero16(f32x4v* acc, int n) {
    for (int t = 0; t < n; ++t) acc[t] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
}

}  // namespace nsa
