// render_sdfnet4.hip -- the per-point SDF network kernels of the composite pass in the QUAD tiling (mlp16.hpp, sdf_net4.hpp):
// a wave = 16 points, four lanes per point, v_mfma_f32_16x16x32_bf16.  Same mathematics, entry points and buffers as
// render_sdfnet.hip (the 32-point tiling, whose header comment derives the forward / reverse pass / tangent sweep / reverse
// sweep); selected with nsa_grid_t.tile == 16 and a packed block from fused/pack.py::pack_sdf_net4.
//   k_sdfnet4_fwd   sdf, 64-feature vector (HL layout, shared with the colour kernels) and grad sdf of one network
//   k_sdfnet4_bwd   value path + double backward through the reverse pass (grid-Hessian term dropped, hashgrid.py:134);
//                   MAP = true adds table gradients (run-merged atomics) and the emission rows of the weight gradients
// Reference: ImplicitNetworkGrid.get_outputs/gradient (code/model/base_networks.py:195-221), ImplicitNetworkGrid_COMBINE (:7-47).
#include <cstdlib>
#include "sdf_net4.hpp"

namespace nsa {

// Waves per workgroup, per kernel.  8 waves: 128 points share one staged copy of every weight block, 36-KiB stage buffers (every
// block in one piece), 123 KB of LDS = one workgroup per CU.  4 waves: 24-KiB buffers -- the two first-layer blocks arrive in two
// parts (mlp16.hpp) -- and 2 x 24 KiB + 24 KiB of grid Jacobians = 74 KB, so TWO workgroups (two independent barrier domains)
// share a CU.  Measured (profiles/r02_ab_experiments.txt r3j): the forward is 3.5 % faster with 4 waves, the backward 7 % slower
// (two more barriers per tile, 7 spilled registers) -- hence the split default.
#ifndef NSA_NW4_FWD
#define NSA_NW4_FWD 4
#endif
#ifndef NSA_NW4_BWD
#define NSA_NW4_BWD 8
#endif
// the paired forward (both networks, ten staged GEMMs per tile) stages twice the weights of a single forward: 8-wave workgroups
// measured 106.3 -> 102.9 us against 4-wave ones (profiles/r03_ab_experiments.txt r3af)
#ifndef NSA_NW4_PAIR
#define NSA_NW4_PAIR 8
#endif
constexpr int stage_floats4(int nw) { return nw >= 8 ? 9216 : 6144; }

struct SdfNet4Args {
    PointSrc src;
    const float* table;
    const float* wp;
    float divide_factor;
    int accumulate;        // 0: overwrite outputs, 1: add to them (second network of the COMBINE)
    float* sdf;            // [P]
    float* grad;           // [P,3]
    float* feat;           // HL [ceil(P/32)*32*64]
    const float* g_sdf;    // [P]
    const float* g_feat;   // HL
    const float* g_grad;   // [P,3]
    float* g_x;            // [P,3]
    float* g_table;        // table gradient (atomically accumulated) or nullptr
    float* emit;           // emission rows [SE4<NH>::ROWS][emit_ld] or nullptr
    uint32_t emit_ld;
};

// Emission rows (weight-gradient GEMMs; formulas: struct SE of render_sdfnet.hip).  H0 / TIN rows are first-layer slots,
// row = 4 * slot + quarter (96 per region); the other regions are hidden features in reference order (64 rows each):
//   [H0 | TIN | DA_1.. | H_1.. | TH_1..TH_{NH-1} | AB_1..AB_NH | TH_NH | FB]        NH = 1: 512 rows, NH = 3: 1024.
template <int NH>
struct SE4 {
    static constexpr int H0 = 0, TIN = 96;
    __host__ __device__ static constexpr int DA(int k) { return 192 + 64 * (k - 1); }
    __host__ __device__ static constexpr int H(int k) { return 192 + 64 * NH + 64 * (k - 1); }
    __host__ __device__ static constexpr int AB(int k) { return 192 + 128 * NH + 64 * (NH - 1) + 64 * (k - 1); }
    __host__ __device__ static constexpr int TH(int k) { return k < NH ? 192 + 128 * NH + 64 * (k - 1) : AB(1) + 64 * NH; }
    static constexpr int FB = 192 + 128 * NH + 64 * (NH - 1) + 64 * NH + 64;
    static constexpr int ROWS = FB + 64;
};
static_assert(SE4<1>::ROWS == 512 && SE4<3>::ROWS == 1024, "emission row map");

struct Emitter4 {
    float* base;           // emit + column of this lane's point
    uint32_t ld;
    bool live;
    __device__ __forceinline__ void slot(int region, int s, int q, float v) const {
        base[(size_t)(region + 4 * s + q) * ld] = live ? v : 0.0f;
    }
    __device__ __forceinline__ void hid(int region, int s, int q, float v) const {
        base[(size_t)(region + 16 * (s >> 2) + 4 * q + (s & 3)) * ld] = live ? v : 0.0f;
    }
};

// GEMM sequences (block-cooperative weight staging):
//   forward : W0, W_1..W_{NH-1}, WFEAT | reverse pass W_{NH-1}^T..W_1^T, W0^T                                   2 NH + 1
//   backward: W0, W_k | W_k^T.., W0^T | tangent W0, W_k | WFEAT^T | reverse sweep W_k^T.., W0^T                  4 NH + 1
template <int NH, bool BWD>
struct SdfOps4 {
    using P = SdfPack4<NH>;
    static constexpr int NW = BWD ? NSA_NW4_BWD : NSA_NW4_FWD;        // waves per workgroup
    static constexpr int BUF = stage_floats4(NW);                      // stage buffer, floats
    static constexpr int n = BWD ? 4 * NH + 1 : 2 * NH + 1;
    __host__ __device__ static constexpr int rev(int j) { return j < NH - 1 ? P::wht(NH - 1 - j) : P::kW0T; }
    __host__ __device__ static constexpr int fwd(int j) { return j == 0 ? P::kW0 : P::wh(j); }
    __host__ __device__ static constexpr int off(int i) {
        if (!BWD) return i < NH ? fwd(i) : i == NH ? P::kWFEAT : rev(i - NH - 1);
        return i < NH ? fwd(i) : i < 2 * NH ? rev(i - NH) : i < 3 * NH ? fwd(i - 2 * NH) : i == 3 * NH ? P::kWFEATT : rev(i - 3 * NH - 1);
    }
    __host__ __device__ static constexpr int mt(int i) { return off(i) == P::kW0T ? 6 : 4; }
    __host__ __device__ static constexpr int kg(int i) { return off(i) == P::kW0 ? QIN_G : 2; }
    __host__ __device__ static constexpr int n_parts() { return first_part<SdfOps4<NH, BWD>, BUF>(n); }
};

// `wp`: this network's packed block (per-feature vectors).  A sequence over TWO networks (SdfOpsPair) stages from `s0` / `s1`, the
// blocks of its net 0 / net 1; a one-network sequence leaves them null and stages from `wp`.
// Forward of BOTH networks of the COMBINE in one kernel: the coarse network's ops, then the fine network's, one staging chain
// (the last coarse GEMM prefetches the fine network's first block).  off() is relative to the packed block of net(i).
template <int NHA, int NHB>
struct SdfOpsPair {
    using A = SdfOps4<NHA, false>;
    using B = SdfOps4<NHB, false>;
    static constexpr int NW = NSA_NW4_PAIR;
    static constexpr int BUF = stage_floats4(NW);
    static constexpr int n = A::n + B::n;
    __host__ __device__ static constexpr int net(int i) { return i < A::n ? 0 : 1; }
    __host__ __device__ static constexpr int off(int i) { return i < A::n ? A::off(i) : B::off(i - A::n); }
    __host__ __device__ static constexpr int mt(int i) { return i < A::n ? A::mt(i) : B::mt(i - A::n); }
    __host__ __device__ static constexpr int kg(int i) { return i < A::n ? A::kg(i) : B::kg(i - A::n); }
};

template <int NH, class Seq>
__device__ __forceinline__ void hidden_forward4(float* stage, int op0, const float* __restrict__ wp, int lane, int q,
                                                const float (&in)[QIN], float (&sg)[NH][QHS], float (&hlast)[QHS],
                                                const Emitter4* em = nullptr, const float* __restrict__ s0 = nullptr,
                                                const float* __restrict__ s1 = nullptr) {
    using P = SdfPack4<NH>;
    if (!s0) s0 = wp;
    f32x4v acc[4];
    load_vec16(wp + P::kB0, q, acc);
    gemm16_staged<Seq, Seq::NW, Seq::BUF, QIN_G, 4>(stage, s0, op0, lane, in, acc, s1);
#pragma unroll
    for (int k = 1; k <= NH; ++k) {
        float d2;
        const float bound = acc_abs_max16<4>(acc) + kSoftplusSlack;      // (form 2: the next GEMM's scale, known before its operands)
#pragma unroll
        for (int s = 0; s < QHS; ++s) softplus100_all(acc[s >> 2][s & 3], hlast[s], sg[k - 1][s], d2);
        if (em) {
#pragma unroll
            for (int s = 0; s < QHS; ++s) em->hid(SE4<NH>::H(k), s, q, hlast[s]);
        }
        if (k < NH) {
            load_vec16(wp + P::bh(k), q, acc);
            gemm16_staged<Seq, Seq::NW, Seq::BUF, 2, 4>(stage, s0, op0 + k, lane, hlast, acc, s1, &bound);
        }
    }
}

// reverse pass from the sdf output: fills dh[k-1] = dh_k for k = 1..NH-1 (dh_NH is the packed sdf row) and dl = dh_0.
template <int NH, class Seq>
__device__ __forceinline__ void reverse_pass4(float* stage, int op0, const float* __restrict__ wp, int lane, int q,
                                              const float (&sg)[NH][QHS], float (&dh)[NH > 1 ? NH - 1 : 1][QHS], float (&dl)[QIN],
                                              const Emitter4* em = nullptr, const float* __restrict__ s0 = nullptr,
                                              const float* __restrict__ s1 = nullptr) {
    using P = SdfPack4<NH>;
    if (!s0) s0 = wp;
    f32x4v ws[4];
    load_vec16(wp + P::kWSDF, q, ws);
    float da[QHS];
#pragma unroll
    for (int s = 0; s < QHS; ++s) da[s] = sg[NH - 1][s] * ws[s >> 2][s & 3];
    if (em) {
#pragma unroll
        for (int s = 0; s < QHS; ++s) em->hid(SE4<NH>::DA(NH), s, q, da[s]);
    }
    float bound = 0.0f;                                  // (form 2) scale hints: see mlp_common.hpp::acc_abs_max
    const float* hint = nullptr;
#pragma unroll
    for (int k = NH - 1; k >= 1; --k) {
        f32x4v acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
        gemm16_staged<Seq, Seq::NW, Seq::BUF, 2, 4>(stage, s0, op0 + (NH - 1 - k), lane, da, acc, s1, hint);
        bound = acc_abs_max16<4>(acc);                   // |da| <= |acc|: the Softplus derivative is a sigmoid
        hint = &bound;
#pragma unroll
        for (int s = 0; s < QHS; ++s) {
            dh[k - 1][s] = acc[s >> 2][s & 3];
            da[s] = sg[k - 1][s] * acc[s >> 2][s & 3];
        }
        if (em) {
#pragma unroll
            for (int s = 0; s < QHS; ++s) em->hid(SE4<NH>::DA(k), s, q, da[s]);
        }
    }
    f32x4v a6[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) a6[t] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
    gemm16_staged<Seq, Seq::NW, Seq::BUF, 2, 6>(stage, s0, op0 + NH - 1, lane, da, a6, s1, hint);
#pragma unroll
    for (int s = 0; s < QIN; ++s) dl[s] = a6[s >> 2][s & 3];
}

#ifndef NSA_OCC4_FWD
#define NSA_OCC4_FWD 2
#endif
#ifndef NSA_OCC4_BWD
#define NSA_OCC4_BWD 2
#endif

#ifdef NSA_X_TS
static __device__ unsigned long long* g_ts4 = nullptr;
#define TS_BEGIN const unsigned long long ts_start = ts_now(); unsigned long long ts_prev = ts_start; \
    if ((threadIdx.x & 63) == 0) for (int i = 0; i < 16; ++i) nsa_ts_lds[threadIdx.x >> 6][i] = 0;
#define TS_MARK(slot) { const unsigned long long t_ = ts_now(); ts_add(slot, t_ - ts_prev); ts_prev = t_; }
#define TS_END { ts_add(15, ts_now() - ts_start); if (g_ts4 && (threadIdx.x & 63) == 0) { \
    unsigned long long* o_ = g_ts4 + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16; \
    for (int i = 0; i < 16; ++i) o_[i] = nsa_ts_lds[threadIdx.x >> 6][i]; } }
#else
#define TS_BEGIN
#define TS_MARK(slot)
#define TS_END
#endif
template <int L, int C, int NH>
__global__ __launch_bounds__(64 * NSA_NW4_FWD, NSA_OCC4_FWD) void k_sdfnet4_fwd(SdfNet4Args a, GridGeom16 geom) {
    using P = SdfPack4<NH>;
    using Seq = SdfOps4<NH, false>;
    TS_BEGIN
    __shared__ __attribute__((aligned(16))) float stage[2 * Seq::BUF];
    __shared__ LevelGeom s_geom[16];
    stage16_begin<Seq, Seq::NW, Seq::BUF>(stage, a.wp);
    geom_to_lds(geom, s_geom);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    uint32_t tile = blockIdx.x * Seq::NW + (threadIdx.x >> 6);
    const uint32_t n_tiles = (a.src.P + 15) / 16;
    const bool wave_live = tile < n_tiles;                  // a wave without points still takes part in the barriers
    if (!wave_live) tile = n_tiles - 1;
    uint32_t pid = tile * 16 + j;
    const bool live = wave_live && pid < a.src.P;
    if (pid >= a.src.P) pid = a.src.P - 1;
    const uint32_t pt = point_of(a.src, pid);               // point handled by this lane quad
    float x[3], z;
    uint32_t ray;
    load_point(a.src, pt, x, ray, z);
    __syncthreads();                                         // s_geom
    TS_MARK(4)

    // the grid Jacobian of this lane's levels stays in lane-private LDS (24 floats per lane): grad sdf needs no second corner
    // gather at the end of the kernel
    constexpr int kJac = (8 / C) * 3 * C;
    __shared__ float jac_lds[Seq::NW * kJac * 64];
    float* jstore = jac_lds + (threadIdx.x >> 6) * (kJac * 64) + lane;
    // Second network of the COMBINE (accumulate): what it adds to -- the first network's 16 features of this lane, sdf and grad
    // sdf -- is requested NOW, not where it is consumed after the MLP: the workgroup barriers of the staged GEMMs are fences the
    // compiler never moves a load across, and at two waves per SIMD nothing else hides a 1-2 us HBM round trip in mid-kernel.
    float facc[QHS], sacc = 0.0f, gacc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < QHS; ++s) facc[s] = 0.0f;
    if (a.accumulate) {
        const float* fsrc = a.feat + hl_base4(tile, j, q);
#pragma unroll
        for (int s = 0; s < QHS; ++s) facc[s] = fsrc[hl_step4(s)];
        sacc = a.sdf[pt];
#pragma unroll
        for (int d = 0; d < 3; ++d) gacc[d] = a.grad[(size_t)pt * 3 + d];
    }
    float in[QIN];
    pe_slots4(x, q, in);
    grid_slots4<L, C>(x, a.divide_factor, a.table, s_geom, q, in, jstore);
    TS_MARK(5)
    float sg[NH][QHS], hl[QHS];
    hidden_forward4<NH, Seq>(stage, 0, a.wp, lane, q, in, sg, hl);
    TS_MARK(6)
    // outputs: sdf (row 0, VALU dot) and the 64 features (rows 1..64)
    f32x4v ws[4], fo[4];
    load_vec16(a.wp + P::kWSDF, q, ws);
    float part = 0.0f;
#pragma unroll
    for (int s = 0; s < QHS; ++s) part = fmaf(hl[s], ws[s >> 2][s & 3], part);
    float sdf = quad_sum(part) + a.wp[P::kBSDF];
    load_vec16(a.wp + P::kBFEAT, q, fo);
    gemm16_staged<Seq, Seq::NW, Seq::BUF, 2, 4>(stage, a.wp, NH, lane, hl, fo);
    if (wave_live) {
        float* fdst = a.feat + hl_base4(tile, j, q);
#pragma unroll
        for (int s = 0; s < QHS; ++s) fdst[hl_step4(s)] = fo[s >> 2][s & 3] + facc[s];
    }
    TS_MARK(7)
    // grad sdf
    float dh[NH > 1 ? NH - 1 : 1][QHS], dl[QIN], g[3];
    reverse_pass4<NH, Seq>(stage, NH + 1, a.wp, lane, q, sg, dh, dl);
    TS_MARK(8)
    slots_to_x_jac4<L, C>(a.divide_factor, jstore, q, in, dl, g);
#pragma unroll
    for (int d = 0; d < 3; ++d) g[d] = quad_sum(g[d]);
    if (live && q == 0) {
        a.sdf[pt] = sdf + sacc;
#pragma unroll
        for (int d = 0; d < 3; ++d) a.grad[(size_t)pt * 3 + d] = g[d] + gacc[d];
    }
    TS_MARK(9)
    TS_END
}

// ---- both networks in one pass (ImplicitNetworkGrid_COMBINE.get_outputs, base_networks.py:7-47) ---------------------------------
// What two launches of k_sdfnet4_fwd (coarse, then fine with accumulate) compute, with the point, the positional encoding, the
// level geometry and the outputs handled once and the coarse network's results (16 features per lane, sdf, grad sdf) carried in
// registers instead of through the feature buffer: bit-identical results, one kernel tail instead of two.
struct SdfNet4PairArgs {
    PointSrc src;
    const float* table_c; const float* table_f;
    const float* wp_c; const float* wp_f;
    float df_c, df_f;
    float* sdf; float* grad; float* feat;
};

template <int NH, class Seq>
__device__ __forceinline__ void net_forward4(float* stage, int op0, const float* __restrict__ wp, const float* __restrict__ s0,
                                             const float* __restrict__ s1, int lane, int q, const float (&in)[QIN], float& sdf,
                                             f32x4v (&fo)[4], float (&dl)[QIN]) {
    using P = SdfPack4<NH>;
    float sg[NH][QHS], hl[QHS];
    hidden_forward4<NH, Seq>(stage, op0, wp, lane, q, in, sg, hl, nullptr, s0, s1);
    f32x4v ws[4];
    load_vec16(wp + P::kWSDF, q, ws);
    float part = 0.0f;
#pragma unroll
    for (int s = 0; s < QHS; ++s) part = fmaf(hl[s], ws[s >> 2][s & 3], part);
    sdf = quad_sum(part) + wp[P::kBSDF];
    load_vec16(wp + P::kBFEAT, q, fo);
    gemm16_staged<Seq, Seq::NW, Seq::BUF, 2, 4>(stage, s0, op0 + NH, lane, hl, fo, s1);
    float dh[NH > 1 ? NH - 1 : 1][QHS];
    reverse_pass4<NH, Seq>(stage, op0 + NH + 1, wp, lane, q, sg, dh, dl, nullptr, s0, s1);
}

template <int LC, int CC, int NHC, int LF, int CF, int NHF>
__global__ __launch_bounds__(64 * NSA_NW4_PAIR, NSA_OCC4_FWD) void k_sdfnet4_fwd_pair(SdfNet4PairArgs a, GridGeom16 gc, GridGeom16 gf) {
    using Seq = SdfOpsPair<NHC, NHF>;
    __shared__ __attribute__((aligned(16))) float stage[2 * Seq::BUF];
    __shared__ LevelGeom s_geom[32];
    stage16_begin<Seq, Seq::NW, Seq::BUF>(stage, a.wp_c);
    if (threadIdx.x < 16) s_geom[threadIdx.x] = gc.lv[threadIdx.x];
    else if (threadIdx.x < 32) s_geom[threadIdx.x] = gf.lv[threadIdx.x - 16];
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    uint32_t tile = blockIdx.x * Seq::NW + (threadIdx.x >> 6);
    const uint32_t n_tiles = (a.src.P + 15) / 16;
    const bool wave_live = tile < n_tiles;
    if (!wave_live) tile = n_tiles - 1;
#define NSA_BODY_SYNC __syncthreads();
#include "sdfnet4_pair_body.inc"
#undef NSA_BODY_SYNC
}

template <int L, int C, int NH, bool MAP>
__global__ __launch_bounds__(64 * NSA_NW4_BWD, NSA_OCC4_BWD) void k_sdfnet4_bwd(SdfNet4Args a, GridGeom16 geom) {
    using P = SdfPack4<NH>;
    using Seq = SdfOps4<NH, true>;
    using E = SE4<NH>;
    TS_BEGIN
    __shared__ __attribute__((aligned(16))) float stage[2 * Seq::BUF];
    __shared__ LevelGeom s_geom[16];
    stage16_begin<Seq, Seq::NW, Seq::BUF>(stage, a.wp);
    geom_to_lds(geom, s_geom);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    uint32_t tile = blockIdx.x * Seq::NW + (threadIdx.x >> 6);
    const uint32_t n_tiles = (a.src.P + 15) / 16;
    const bool wave_live = tile < n_tiles;
    if (!wave_live) tile = n_tiles - 1;
#define NSA_BODY_SYNC __syncthreads();
#include "sdfnet4_bwd_body.inc"
#undef NSA_BODY_SYNC
    TS_MARK(10)
    TS_END
}

static_assert(NSA_NW4_BWD * 64 * (2 * 8 + 1) <= stage_floats4(NSA_NW4_BWD), "scatter scratch must fit the idle stage buffer");

// ---- resident-weight persistent forms (bf16-operand build; mlp16.hpp::ResidentSeq) -------------------------------------------------
// One 8-wave workgroup per CU copies every distinct packed block of the sequence into LDS once (single piece per fragment: paired
// forward 96 KiB, fine backward 64 KiB) and its waves loop over their tiles with no barrier and no staging: the statements of a tile
// are the staged kernels' own (sdfnet4_pair_body.inc / sdfnet4_bwd_body.inc), so the results are bit-identical to them.
#if NSA_PIECES == 1
template <int LC, int CC, int NHC, int LF, int CF, int NHF>
__global__ __launch_bounds__(64 * NSA_NW4_PAIR, NSA_OCC4_FWD) void k_sdfnet4_fwd_pair_res(SdfNet4PairArgs a, GridGeom16 gc, GridGeom16 gf) {
    using Seq = ResidentSeq<SdfOpsPair<NHC, NHF>>;
    __shared__ __attribute__((aligned(16))) float stage[res_off<Seq>(Seq::n)];
    __shared__ LevelGeom s_geom[32];
    resident_load<Seq, Seq::NW>(stage, a.wp_c, a.wp_f);
    if (threadIdx.x < 16) s_geom[threadIdx.x] = gc.lv[threadIdx.x];
    else if (threadIdx.x < 32) s_geom[threadIdx.x] = gf.lv[threadIdx.x - 16];
    stage_wait();                                            // vmcnt(0) + workgroup barrier: weights and geometry are in LDS
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const uint32_t n_tiles = (a.src.P + 15) / 16;
    constexpr bool wave_live = true;
    for (uint32_t tile = blockIdx.x * Seq::NW + (threadIdx.x >> 6); tile < n_tiles; tile += gridDim.x * Seq::NW) {
#define NSA_BODY_SYNC
#include "sdfnet4_pair_body.inc"
#undef NSA_BODY_SYNC
    }
}

template <int L, int C, int NH>
__global__ __launch_bounds__(64 * NSA_NW4_BWD, NSA_OCC4_BWD) void k_sdfnet4_bwd_res(SdfNet4Args a, GridGeom16 geom) {
    constexpr bool MAP = false;
    using P = SdfPack4<NH>;
    using Seq = ResidentSeq<SdfOps4<NH, true>>;
    using E = SE4<NH>;
    __shared__ __attribute__((aligned(16))) float stage[res_off<Seq>(Seq::n)];
    __shared__ LevelGeom s_geom[16];
    resident_load<Seq, Seq::NW>(stage, a.wp);
    geom_to_lds(geom, s_geom);
    stage_wait();
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const uint32_t n_tiles = (a.src.P + 15) / 16;
    constexpr bool wave_live = true;
    for (uint32_t tile = blockIdx.x * Seq::NW + (threadIdx.x >> 6); tile < n_tiles; tile += gridDim.x * Seq::NW) {
#define NSA_BODY_SYNC
#pragma push_macro("TS_MARK")             // (the per-phase cycle stamps of the profiling build belong to the staged kernel)
#undef TS_MARK
#define TS_MARK(slot)
#include "sdfnet4_bwd_body.inc"
#pragma pop_macro("TS_MARK")
#undef NSA_BODY_SYNC
    }
}

// workgroups of a persistent launch: one per CU
static int persistent_blocks4() {
    // per device id (a process may drive GPUs with different CU counts, and the current device may change between launches)
    static int n[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (n[dev] == 0) {
        int cus = 0;
        n[dev] = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ? cus : 256;
    }
    return n[dev];
}
// NSA_BF16_RESIDENT=0: the staged forms (A/B runs)
static bool bf16_resident() {
    static const bool on = [] { const char* e = getenv("NSA_BF16_RESIDENT"); return !(e && e[0] == '0'); }();
    return on;
}
#endif

static int launch_sdfnet4(bool bwd, const nsa_grid_t* grid, const SdfNet4Args& a, hipStream_t st) {
    const bool map = a.g_table != nullptr || a.emit != nullptr;
    GridGeom16 geom;
    if (int rc = make_grid_geom16(grid->offsets_host, grid->L, grid->S, grid->H, &geom, grid->C)) return rc;
    const uint32_t tiles = (a.src.P + 15) / 16;
    const int nw = bwd ? NSA_NW4_BWD : NSA_NW4_FWD;
    const dim3 g((tiles + nw - 1) / nw), b(64 * nw);
    launch_begin();
    if (grid->L == 4 && grid->C == 8 && grid->n_hidden == 1) {
        if (bwd && map) hipLaunchKernelGGL((k_sdfnet4_bwd<4, 8, 1, true>), g, b, 0, st, a, geom);
        else if (bwd)   hipLaunchKernelGGL((k_sdfnet4_bwd<4, 8, 1, false>), g, b, 0, st, a, geom);
        else            hipLaunchKernelGGL((k_sdfnet4_fwd<4, 8, 1>), g, b, 0, st, a, geom);
    } else if (grid->L == 8 && grid->C == 4 && grid->n_hidden == 3) {
        if (bwd && map) hipLaunchKernelGGL((k_sdfnet4_bwd<8, 4, 3, true>), g, b, 0, st, a, geom);
#if NSA_PIECES == 1
        else if (bwd && bf16_resident()) {
            const uint32_t wgs = (tiles + nw - 1) / nw, cap = (uint32_t)persistent_blocks4();
            hipLaunchKernelGGL((k_sdfnet4_bwd_res<8, 4, 3>), dim3(wgs < cap ? wgs : cap), b, 0, st, a, geom);
        }
#endif
        else if (bwd)   hipLaunchKernelGGL((k_sdfnet4_bwd<8, 4, 3, false>), g, b, 0, st, a, geom);
        else            hipLaunchKernelGGL((k_sdfnet4_fwd<8, 4, 3>), g, b, 0, st, a, geom);
    } else {
        return NSA_EUNSUPPORTED_NET;
    }
    return launch_end();
}

}  // namespace nsa

#include "quad_entries.hpp"

// Internal entry points (not in the public header): render_sdfnet.hip forwards here when nsa_grid_t.tile == 16.
extern "C" {

int NSA_ENTRY(nsa_sdfnet4_forward)(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, int accumulate, float* sdf,
                                   float* grad, float* feat_hl, nsa_stream_t stream) {
    using namespace nsa;
    SdfNet4Args a{};
    a.src = PointSrc{pts->rays_o, pts->rays_d, pts->z_vals, pts->points, pts->P, pts->S, pts->order};
    a.table = grid->table; a.wp = packed; a.divide_factor = grid->divide_factor; a.accumulate = accumulate;
    a.sdf = sdf; a.grad = grad; a.feat = feat_hl;
    return launch_sdfnet4(false, grid, a, (hipStream_t)stream);
}

int NSA_ENTRY(nsa_sdfnet4_backward)(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, const float* g_sdf,
                                    const float* g_feat_hl, const float* g_grad, int accumulate, float* g_x, float* g_table,
                                    float* emit, uint32_t emit_ld, nsa_stream_t stream) {
    using namespace nsa;
    SdfNet4Args a{};
    a.src = PointSrc{pts->rays_o, pts->rays_d, pts->z_vals, pts->points, pts->P, pts->S, pts->order};
    a.table = grid->table; a.wp = packed; a.divide_factor = grid->divide_factor; a.accumulate = accumulate;
    a.g_sdf = g_sdf; a.g_feat = g_feat_hl; a.g_grad = g_grad; a.g_x = g_x;
    a.g_table = g_table; a.emit = emit; a.emit_ld = emit_ld;
    return launch_sdfnet4(true, grid, a, (hipStream_t)stream);
}

int NSA_ENTRY(nsa_sdfnet4_forward_pair)(const nsa_points_t* pts, const nsa_grid_t* coarse, const nsa_grid_t* fine,
                                        const float* packed_coarse, const float* packed_fine, float* sdf, float* grad,
                                        float* feat_hl, nsa_stream_t stream) {
    using namespace nsa;
    if (!(coarse->L == 4 && coarse->C == 8 && coarse->n_hidden == 1 && fine->L == 8 && fine->C == 4 && fine->n_hidden == 3))
        return NSA_EUNSUPPORTED_NET;
    SdfNet4PairArgs a{};
    a.src = PointSrc{pts->rays_o, pts->rays_d, pts->z_vals, pts->points, pts->P, pts->S, pts->order};
    a.table_c = coarse->table; a.table_f = fine->table; a.wp_c = packed_coarse; a.wp_f = packed_fine;
    a.df_c = coarse->divide_factor; a.df_f = fine->divide_factor;
    a.sdf = sdf; a.grad = grad; a.feat = feat_hl;
    GridGeom16 gc, gf;
    if (int rc = make_grid_geom16(coarse->offsets_host, coarse->L, coarse->S, coarse->H, &gc, coarse->C)) return rc;
    if (int rc = make_grid_geom16(fine->offsets_host, fine->L, fine->S, fine->H, &gf, fine->C)) return rc;
    const uint32_t tiles = (a.src.P + 15) / 16;
    launch_begin();
#if NSA_PIECES == 1
    if (bf16_resident()) {
        const uint32_t wgs = (tiles + NSA_NW4_PAIR - 1) / NSA_NW4_PAIR, cap = (uint32_t)persistent_blocks4();
        hipLaunchKernelGGL((k_sdfnet4_fwd_pair_res<4, 8, 1, 8, 4, 3>), dim3(wgs < cap ? wgs : cap), dim3(64 * NSA_NW4_PAIR), 0,
                           (hipStream_t)stream, a, gc, gf);
        return launch_end();
    }
#endif
    hipLaunchKernelGGL((k_sdfnet4_fwd_pair<4, 8, 1, 8, 4, 3>), dim3((tiles + NSA_NW4_PAIR - 1) / NSA_NW4_PAIR), dim3(64 * NSA_NW4_PAIR), 0,
                       (hipStream_t)stream, a, gc, gf);
    return launch_end();
}

#ifdef NSA_X_TS
int NSA_ENTRY(nsa_debug_set_ts)(unsigned long long* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(nsa::g_ts4), &p, sizeof(p)) == hipSuccess ? 0 : 3;
}
#endif

int NSA_ENTRY(nsa_sdfnet4_emit_rows)(uint32_t n_hidden) {
    return n_hidden == 1 ? nsa::SE4<1>::ROWS : n_hidden == 3 ? nsa::SE4<3>::ROWS : -1;
}

}  // extern "C"
