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

// the split pieces of one k-group's B operand (split once per k-group, shared by all output tiles); unused members cost nothing
struct B16 {
    bf16x8_t bh, bm, bl;      // form 3 (bf16 build: bh only)
    f16x8_t h0, h1;           // form 2
};
__device__ __forceinline__ void split_b16(const float (&x)[8], B16& o) {
    if constexpr (kPieces == 3) {
        BFrag bf;
        split8(x, bf);
        o.bh = as_bf16x8(bf.p[0]); o.bm = as_bf16x8(bf.p[1]); o.bl = as_bf16x8(bf.p[2]);
    } else if constexpr (kPieces == 2) {
        split8_h2(x, o.h0, o.h1);
    } else {
        o.bh = round8_bf16(x);
    }
}

// acc[t] += A(t, group) * x for TC output tiles of one k-group: the six cross products of the 3-way split, the four of the two-piece
// fp16 form, or one bf16 product
template <int TC>
__device__ __forceinline__ void mma16_tiles(const u32x4 (&a)[TC][3], const B16& b, f32x4v* acc) {
    if constexpr (kPieces == 3) {
#define NSA_MM16(AP, BV)                                                                               \
        _Pragma("unroll") for (int t = 0; t < TC; ++t)                                                 \
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a[t][AP]), BV, acc[t], 0, 0, 0);
        NSA_MM16(2, b.bh) NSA_MM16(0, b.bl) NSA_MM16(1, b.bm) NSA_MM16(1, b.bh) NSA_MM16(0, b.bm) NSA_MM16(0, b.bh)
#undef NSA_MM16
    } else if constexpr (kPieces == 2) {
#define NSA_MM16H(AP, BV)                                                                              \
        _Pragma("unroll") for (int t = 0; t < TC; ++t)                                                 \
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(a[t][AP]), BV, acc[t], 0, 0, 0);
#if NSA_FORM2_PRODUCTS == 4
        NSA_MM16H(1, b.h1)
#endif
        NSA_MM16H(0, b.h1) NSA_MM16H(1, b.h0) NSA_MM16H(0, b.h0)
#undef NSA_MM16H
    } else {
#pragma unroll
        for (int t = 0; t < TC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a[t][0]), b.bh, acc[t], 0, 0, 0);
    }
}

// quad tiling: a point's operands live in lanes j, j + 16, j + 32, j + 48
template <int N>
__device__ __forceinline__ PointScale point_scale16(const float (&b)[N], const float* hint = nullptr) {
    const float m = hint ? *hint : abs_max<N>(b);
    return point_scale_of(__uint_as_float(umax_xor32(umax_xor16(__float_as_uint(m)))));
}
// this lane's largest |accumulator| (see mlp_common.hpp::acc_abs_max: the bound a following GEMM can take its scale from)
template <int MT>
__device__ __forceinline__ float acc_abs_max16(const f32x4v (&acc)[MT]) {
    float m[2] = {0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        m[0] = fmaxf(fmaxf(m[0], fabsf(acc[t][0])), fabsf(acc[t][1]));
        m[1] = fmaxf(fmaxf(m[1], fabsf(acc[t][2])), fabsf(acc[t][3]));
    }
    return fmaxf(m[0], m[1]);
}
template <int MT, bool POST>
__device__ __forceinline__ void scale_acc16(f32x4v (&acc)[MT], int k) {
#ifdef NSA_X_NO_ACC_SCALE
    return;
#endif
#ifdef NSA_X_NO_PRE
    if (!POST) return;
#endif
#ifdef NSA_X_NO_POST
    if (POST) return;
#endif
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = __builtin_ldexpf(acc[t][r], k);
}

// acc[MT] += A[MT x KG] * b, A block in LDS ([mt][g][piece][lane]), b = this lane's 8*KG k-values.
// Output tiles are processed in chunks of at most 4 (4 independent accumulators per product step: no MFMA waits on its
// predecessor; 12 fragment registers live per chunk instead of 3*MT).
template <int KG, int MT>
__device__ __forceinline__ void gemm16_lds(const float* lds_block, int lane, const float (&b)[8 * KG], f32x4v (&acc)[MT],
                                           const float* hint = nullptr) {
    const lds_u4* w4 = (const lds_u4*)lds_block + lane;
    constexpr int TC = MT <= 4 ? MT : (MT % 3 == 0 ? 3 : 4);
    static_assert(MT % TC == 0, "tile chunking");
    PointScale ps{1.0f, 0};
    if constexpr (kPieces == 2) {
        ps = point_scale16<8 * KG>(b, hint);
        scale_acc16<MT, false>(acc, ps.kpre);
    }
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = kPieces == 2 ? b[8 * g + e] * ps.s : b[8 * g + e];
        B16 bp;
        split_b16(x, bp);
#pragma unroll
        for (int c = 0; c < MT / TC; ++c) {
            u32x4 a[TC][3];
#pragma unroll
            for (int t = 0; t < TC; ++t)
#pragma unroll
                for (int pc = 0; pc < kPieces; ++pc) a[t][pc] = w4[(((c * TC + t) * KG + g) * kLdsPieces + pc) * 64];
#ifdef NSA_X_LDS_FENCE       // SLP-hazard bisect: every fragment read has RETURNED before the first MFMA of the chunk issues
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#endif
            mma16_tiles<TC>(a, bp, &acc[c * TC]);
        }
    }
    if constexpr (kPieces == 2) scale_acc16<MT, true>(acc, -ps.kpre);
    __builtin_amdgcn_sched_barrier(0);   // keep later layers' LDS reads from being hoisted above this GEMM
}

// the same from global memory (per-wave streaming of the packed block; sampler fallback)
template <int KG, int MT>
__device__ __forceinline__ void gemm16_glb(const float* __restrict__ wp, int lane, const float (&b)[8 * KG], f32x4v (&acc)[MT],
                                           const float* hint = nullptr) {
    const uint4* __restrict__ w4 = reinterpret_cast<const uint4*>(wp) + lane;
    constexpr int TC = MT <= 4 ? MT : (MT % 3 == 0 ? 3 : 4);
    PointScale ps{1.0f, 0};
    if constexpr (kPieces == 2) {
        ps = point_scale16<8 * KG>(b, hint);
        scale_acc16<MT, false>(acc, ps.kpre);
    }
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = kPieces == 2 ? b[8 * g + e] * ps.s : b[8 * g + e];
        B16 bp;
        split_b16(x, bp);
#pragma unroll
        for (int c = 0; c < MT / TC; ++c) {
            u32x4 a[TC][3];
#pragma unroll
            for (int t = 0; t < TC; ++t)
#pragma unroll
                for (int pc = 0; pc < kPieces; ++pc) {
                    const uint4 v = w4[(((c * TC + t) * KG + g) * 3 + kSlot0 + pc) * 64];
                    a[t][pc] = u32x4{v.x, v.y, v.z, v.w};
                }
            mma16_tiles<TC>(a, bp, &acc[c * TC]);
        }
    }
    if constexpr (kPieces == 2) scale_acc16<MT, true>(acc, -ps.kpre);
    __builtin_amdgcn_sched_barrier(0);
}

// ---- block-cooperative weight staging for NW waves per workgroup (see mlp_common.hpp for the 4-wave original) -------------
template <int NW>
__device__ __forceinline__ void stage_issue_n(const float* __restrict__ g, int nfloats, float* lds_dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = nfloats / 256 / 3 * kLdsPieces;      // (mlp_common.hpp::kLdsPieces / src_frag: the pieces this build multiplies with)
    for (int ch = wave; ch < chunks; ch += NW)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + src_frag(ch) * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(lds_dst + ch * 256), 16, 0, 0);
}

// ---- staged GEMMs, part by part --------------------------------------------------------------------------------------------
// A kernel's GEMM sequence Seq lists logical ops (Seq::n; Seq::off(i) = packed block, Seq::mt(i) output tiles, Seq::kg(i)
// k-groups).  A block larger than the stage buffer (BUF floats) is staged in PARTS of whole k-groups -- the part "groups
// g0 .. g0+ng-1 of every tile" is MT strided segments in global memory, laid out in LDS as [mt][ng][piece][lane] -- so that the
// buffers can be small enough for two workgroups per CU.  Parts are numbered through the whole sequence; part p uses buffer p & 1
// and fetches part p + 1 while it multiplies.  With BUF >= the largest block every op is one part (the round-2 layout).
struct StagePart {
    int off, mt, kg, g0, ng;
    int net;      // 0 / 1: which of the (at most two) packed blocks of the sequence `off` is relative to (Seq::net, optional)
};

// Seq::net(op) is optional: sequences over ONE packed block do not define it
template <class Seq, class = void>
struct seq_has_net { static constexpr bool value = false; };
template <class Seq>
struct seq_has_net<Seq, decltype((void)Seq::net(0))> { static constexpr bool value = true; };
template <class Seq>
__host__ __device__ constexpr int seq_net(int op) {
    if constexpr (seq_has_net<Seq>::value) return Seq::net(op);
    else return 0;
}

template <int BUF>
__host__ __device__ constexpr int max_groups(int mt) { return BUF / (mt * 768) < 1 ? 1 : BUF / (mt * 768); }

template <class Seq, int BUF>
__host__ __device__ constexpr int parts_of(int op) {
    const int m = max_groups<BUF>(Seq::mt(op));
    return (Seq::kg(op) + m - 1) / m;
}

template <class Seq, int BUF>
__host__ __device__ constexpr int first_part(int op) {
    int p = 0;
    for (int i = 0; i < op; ++i) p += parts_of<Seq, BUF>(i);
    return p;
}

template <class Seq, int BUF>
__host__ __device__ constexpr StagePart part_at(int p) {
    int op = 0;
    while (op < Seq::n && p >= parts_of<Seq, BUF>(op)) { p -= parts_of<Seq, BUF>(op); ++op; }
    if (op >= Seq::n) return StagePart{0, 0, 0, 0, 0, 0};
    const int m = max_groups<BUF>(Seq::mt(op));
    const int g0 = p * m;
    const int ng = Seq::kg(op) - g0 < m ? Seq::kg(op) - g0 : m;
    return StagePart{Seq::off(op), Seq::mt(op), Seq::kg(op), g0, ng, seq_net<Seq>(op)};
}

template <int NW>
__device__ __forceinline__ void stage_issue_part(const float* __restrict__ wp, const StagePart o, float* lds_dst,
                                                 const float* __restrict__ wp1 = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_tile = o.ng * kLdsPieces;        // 1 KiB chunks per tile in this part
    const int chunks = o.mt * per_tile;
    const float* base = o.net ? wp1 : wp;
    for (int ch = wave; ch < chunks; ch += NW) {
        const int mt = ch / per_tile, rem = ch - mt * per_tile;
        const float* src = base + o.off + ((mt * o.kg + o.g0) * 3 + src_frag(rem)) * 256 + lane * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds_dst + ch * 256), 16, 0, 0);
    }
}

template <class Seq, int NW, int BUF>
__device__ __forceinline__ void stage16_begin(float* stage, const float* __restrict__ wp) {
    stage_issue_part<NW>(wp, part_at<Seq, BUF>(0), stage);
}


// groups G0 .. G0+NG-1 of a KG-group GEMM from a staged part laid out [mt][NG][piece][lane]
template <int KG, int MT, int G0, int NG, bool PRE = true, bool POST = true>      // PRE / POST: see mlp_common.hpp::gemm_lds_part
__device__ __forceinline__ void gemm16_lds_part(const float* lds_block, int lane, const float (&b)[8 * KG], f32x4v (&acc)[MT],
                                                const float* hint = nullptr) {
    const lds_u4* w4 = (const lds_u4*)lds_block + lane;
    constexpr int TC = MT <= 4 ? MT : (MT % 3 == 0 ? 3 : 4);
    static_assert(MT % TC == 0, "tile chunking");
    PointScale ps{1.0f, 0};
    if constexpr (kPieces == 2) {           // (every part of a GEMM sees the same b, hence the same scale)
        ps = point_scale16<8 * KG>(b, hint);
        if constexpr (PRE) scale_acc16<MT, false>(acc, ps.kpre);
    }
#pragma unroll
    for (int gl = 0; gl < NG; ++gl) {
        const int g = G0 + gl;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = kPieces == 2 ? b[8 * g + e] * ps.s : b[8 * g + e];
        B16 bp;
        split_b16(x, bp);
#pragma unroll
        for (int c = 0; c < MT / TC; ++c) {
            u32x4 a[TC][3];
#pragma unroll
            for (int t = 0; t < TC; ++t)
#pragma unroll
                for (int pc = 0; pc < kPieces; ++pc) a[t][pc] = w4[(((c * TC + t) * NG + gl) * kLdsPieces + pc) * 64];
            mma16_tiles<TC>(a, bp, &acc[c * TC]);
        }
    }
    if constexpr (kPieces == 2 && POST) scale_acc16<MT, true>(acc, -ps.kpre);
    __builtin_amdgcn_sched_barrier(0);   // keep later layers' LDS reads from being hoisted above this GEMM
}

template <class Seq, int NW, int BUF, int KG, int MT, int P0, int G0, int NG, bool PRE = true, bool POST = true>
__device__ __forceinline__ void gemm16_staged_part(float* stage, const float* __restrict__ wp, int part, int lane,
                                                   const float (&b)[8 * KG], f32x4v (&acc)[MT], const float* __restrict__ wp1 = nullptr,
                                                   const float* hint = nullptr) {
    stage_wait();
    const StagePart nxt = part_at<Seq, BUF>(part + 1);
    if (nxt.mt) stage_issue_part<NW>(wp, nxt, stage + ((part + 1) & 1) * BUF, wp1);

#ifdef NSA_X_TS
    const unsigned long long tg = ts_now();
#endif
    gemm16_lds_part<KG, MT, G0, NG, PRE, POST>(stage + (part & 1) * BUF, lane, b, acc, hint);
#ifdef NSA_X_TS
    ts_add(3, ts_now() - tg);
#endif
}

// ---- resident weights (bf16-operand build, round 5) ---------------------------------------------------------------------------
// With one piece per fragment the packed blocks of a whole kernel fit into LDS at once (paired forward: 96 KiB, fine backward: 64 KiB):
// a persistent workgroup copies every DISTINCT block of its sequence once, and its waves then loop over their tiles with no barrier
// and no staging traffic at all -- the waves of a SIMD drift apart instead of meeting at a barrier per GEMM.  ResidentSeq<Seq> marks
// a sequence as resident: gemm16_staged then multiplies straight out of the resident image (blocks laid out [mt][kg][piece][lane]).
template <class Seq>
struct ResidentSeq : Seq { static constexpr bool kResident = true; };
template <class Seq, class = void>
struct seq_resident { static constexpr bool value = false; };
template <class Seq>
struct seq_resident<Seq, decltype((void)Seq::kResident)> { static constexpr bool value = true; };

// the first op of the sequence that uses the same packed block as `op` (a backward names its forward and reverse blocks twice)
template <class Seq>
__host__ __device__ constexpr int res_first(int op) {
    for (int j = 0; j < op; ++j)
        if (Seq::off(j) == Seq::off(op) && seq_net<Seq>(j) == seq_net<Seq>(op)) return j;
    return op;
}
template <class Seq>
__host__ __device__ constexpr int res_size(int op) { return Seq::mt(op) * Seq::kg(op) * 256 * kLdsPieces; }
// LDS offset (floats) of op's block in the resident image; op == Seq::n: the image's size
template <class Seq>
__host__ __device__ constexpr int res_off(int op) {
    const int f = op < Seq::n ? res_first<Seq>(op) : op;
    int o = 0;
    for (int j = 0; j < f; ++j)
        if (res_first<Seq>(j) == j) o += res_size<Seq>(j);
    return o;
}
template <class Seq, int NW>
__device__ __forceinline__ void resident_load(float* lds, const float* __restrict__ wp, const float* __restrict__ wp1 = nullptr) {
#pragma unroll
    for (int op = 0; op < Seq::n; ++op)
        if (res_first<Seq>(op) == op)
            stage_issue_part<NW>(wp, StagePart{Seq::off(op), Seq::mt(op), Seq::kg(op), 0, Seq::kg(op), seq_net<Seq>(op)},
                                 lds + res_off<Seq>(op), wp1);
}

// (Round 6 built a barrier-free form of the staged fp32 kernels -- a ring of three stage buffers with per-part arrival counters in LDS
// instead of the s_barrier per GEMM, RingSeq / gemm16_ring / k_sdfnet4_{fwd_pair,bwd}_ring -- bit-identical to the staged kernels and
// 3-10 % SLOWER: the waves did not start to co-execute (matrix / vector co-execution 0.25 -> 0.28 of the matrix time, 5 % more vector
// instructions for the hand-off).  profiles/r06_ab_experiments.txt r6a / r6b, profiles/r06_quad_ring_pmc_*.csv; last present in commit
// b5f9d27.)

// logical GEMM `opi` of Seq (KG k-groups, MT output tiles): all its parts (at most two).  `wp1`: base of the second packed block
// of a sequence that runs two networks (Seq::net).
template <class Seq, int NW, int BUF, int KG, int MT>
__device__ __forceinline__ void gemm16_staged(float* stage, const float* __restrict__ wp, int opi, int lane,
                                              const float (&b)[8 * KG], f32x4v (&acc)[MT], const float* __restrict__ wp1 = nullptr,
                                              const float* hint = nullptr) {       // hint: acc_abs_max16-style bound on |b| (form 2)
    if constexpr (seq_resident<Seq>::value) {        // `stage` = the resident image: no wait, no copy, no barrier
        gemm16_lds_part<KG, MT, 0, KG>(stage + res_off<Seq>(opi), lane, b, acc, hint);
        return;
    }
    constexpr int M = max_groups<BUF>(MT);
    constexpr int NP = (KG + M - 1) / M;
    static_assert(NP <= 3, "a staged GEMM is split into at most three parts");
    const int p0 = first_part<Seq, BUF>(opi);
    float own = 0.0f;                                // several parts: this lane's maximum once, handed to every part
    if constexpr (kPieces == 2 && NP > 1) {
        if (!hint) {
            own = abs_max<8 * KG>(b);
            hint = &own;
        }
    }
    gemm16_staged_part<Seq, NW, BUF, KG, MT, 0, 0, (KG < M ? KG : M), true, NP == 1>(stage, wp, p0, lane, b, acc, wp1, hint);
    if constexpr (NP > 1) gemm16_staged_part<Seq, NW, BUF, KG, MT, 1, M, (KG - M < M ? KG - M : M), false, NP == 2>(stage, wp, p0 + 1, lane, b, acc, wp1, hint);
    if constexpr (NP > 2) gemm16_staged_part<Seq, NW, BUF, KG, MT, 2, 2 * M, KG - 2 * M, false, true>(stage, wp, p0 + 2, lane, b, acc, wp1, hint);
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
