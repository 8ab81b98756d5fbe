// render_sdfnet.hip -- per-point SDF network kernels of the composite pass:
//   k_sdfnet_fwd   sdf, feature vector and grad sdf (the reverse pass of base_networks.py:214-219) of one network
//   k_sdfnet_bwd   its hand-derived backward: value path + the double-backward through the reverse pass,
//                  with exactly the terms the reference graph has (hash-grid Hessian dropped, hashgrid.py:134)
// Reference: ImplicitNetworkGrid.get_outputs/gradient (code/model/base_networks.py:195-221),
//            ImplicitNetworkGrid_COMBINE (:7-47): the coarse and fine networks are run as two launches that
//            accumulate into the same sdf / grad / feature buffers.
//
// One wave = 32 points (lane pair per point), activations in registers, each layer's packed weights staged once per
// workgroup in LDS (mlp_common.hpp).  Per-point feature vectors travel between kernels in "HL" layout: float index
// ((tile*32 + q)*64 + lane), q = 16 t + r -- i.e. exactly the register image of the MFMA result, so the consumer's
// B operand is a coalesced 256-byte load per register.
//
// Notation (per network, NH hidden layers): h0 = first-layer input slots, a_k = W_{k-1} h_{k-1} + b, h_k = sp(a_k),
// out = W_NH h_NH + b = [sdf, feat];   reverse pass: dh_NH = W_NH[0,:], da_k = sp'(a_k) * dh_k,
// dh_{k-1} = W_{k-1}^T da_k, grad sdf = G(x, dh_0).
// Backward, given (sbar, fbar, nbar) = cotangents of (sdf, feat, grad sdf):
//   tangent sweep   t_0 = dG/d(dh_0)^T nbar ;  ta_k = W_{k-1} th_{k-1} ;  e_k = sp''(a_k) dh_k ta_k ;  th_k = sp'(a_k) ta_k
//   reverse sweep   hb_NH = sbar W_NH[0,:] + Wfeat^T fbar ;  ab_k = sp'(a_k) hb_k + e_k ;  hb_{k-1} = W_{k-1}^T ab_k
//   xbar            = G(x, hb_0) + nbar * (PE second-derivative term)           [no grid-Hessian term]
#include "sdf_net.hpp"

namespace nsa {

struct SdfNetArgs {
    PointSrc src;
    const float* table;
    const float* wp;
    float divide_factor;
    int accumulate;        // 0: overwrite outputs, 1: add to them (second network of the COMBINE)
    // forward outputs
    float* sdf;            // [P]
    float* grad;           // [P,3]
    float* feat;           // HL [tiles*32*64]
    // backward inputs / outputs
    const float* g_sdf;    // [P]
    const float* g_feat;   // HL
    const float* g_grad;   // [P,3]
    float* g_x;            // [P,3]
    // mapping (parameter gradients)
    float* g_table;        // table gradient, same shape as `table` (atomically accumulated) or nullptr
    float* emit;           // per-point vectors for the weight-gradient GEMMs, [SE_ROWS][emit_ld] or nullptr
    uint32_t emit_ld;
};

// Rows of the emission buffer of a network with NH hidden layers; column = tile*32 + point-in-tile.  With
//   AB_k = total cotangent of a_k, DA_k = sp'(a_k) dh_k (reverse pass), H_k = h_k, TH_k = sp'(a_k) ta_k (tangent sweep), FB = fbar:
//   dW_0 = AB_1 H0^T + DA_1 TIN^T,   dW_k = AB_{k+1} H_k^T + DA_{k+1} TH_k^T  (0 < k < NH),   db_k = sum AB_{k+1},
//   dW_NH[0] = sum sbar H_NH + TH_NH,   dW_NH[1:] = FB H_NH^T,   db_NH = [sum sbar, sum FB].
// H0/TIN rows are first-layer slots (row = 2*slot + half), the others are hidden features in reference order.
// Region order [H0 | TIN | DA_1.. | H_1.. | TH_1..TH_{NH-1} | AB_1..AB_NH | TH_NH | FB]: everything whose row sums are
// needed (AB_k, TH_NH, FB) is contiguous -- one reduction.  NH = 1 gives 0/72/144/208/272/336/400, 464 rows; NH = 3: 976.
template <int NH>
struct SE {
    static constexpr int H0 = 0, TIN = 72;
    __host__ __device__ static constexpr int DA(int k) { return 144 + 64 * (k - 1); }
    __host__ __device__ static constexpr int H(int k) { return 144 + 64 * NH + 64 * (k - 1); }
    __host__ __device__ static constexpr int AB(int k) { return 144 + 128 * NH + 64 * (NH - 1) + 64 * (k - 1); }
    __host__ __device__ static constexpr int TH(int k) { return k < NH ? 144 + 128 * NH + 64 * (k - 1) : AB(1) + 64 * NH; }
    static constexpr int FB = 144 + 128 * NH + 64 * (NH - 1) + 64 * NH + 64;
    static constexpr int ROWS = FB + 64;
};
static_assert(SE<1>::DA(1) == 144 && SE<1>::H(1) == 208 && SE<1>::AB(1) == 272 && SE<1>::TH(1) == 336 && SE<1>::FB == 400 &&
              SE<1>::ROWS == 464 && SE<3>::ROWS == 976, "emission row map");

struct Emitter {
    float* base;
    uint32_t ld;
    bool live;
    __device__ __forceinline__ void slot(int region, int s, int h, float v) const {
        base[(size_t)(region + 2 * s + h) * ld] = live ? v : 0.0f;
    }
    __device__ __forceinline__ void hid(int region, int q, int h, float v) const {      // q = 16 t + r
        const int f = 32 * (q >> 4) + (q & 3) + 8 * ((q & 15) >> 2) + 4 * h;
        base[(size_t)(region + f) * ld] = live ? v : 0.0f;
    }
};

// GEMM sequences of the two kernels (block-cooperative weight staging, mlp_common.hpp):
//   forward : W0, W_1..W_{NH-1}, WFEAT | reverse pass W_{NH-1}^T..W_1^T, W0^T                                   2 NH + 1
//   backward: W0, W_k | W_k^T.., W0^T | tangent W0, W_k | WFEAT^T | reverse sweep W_k^T.., W0^T                  4 NH + 1
template <int NH, bool BWD>
struct SdfOps {
    using P = SdfPack<NH>;
    static constexpr int n = BWD ? 4 * NH + 1 : 2 * NH + 1;
    __host__ __device__ static constexpr int rev(int j) { return j < NH - 1 ? P::wht(NH - 1 - j) : P::kW0T; }   // j-th op of a reverse chain
    __host__ __device__ static constexpr int fwd(int j) { return j == 0 ? P::kW0 : P::wh(j); }
    __host__ __device__ static constexpr int off(int i) {
        if (!BWD) return i < NH ? fwd(i) : i == NH ? P::kWFEAT : rev(i - NH - 1);
        return i < NH ? fwd(i) : i < 2 * NH ? rev(i - NH) : i < 3 * NH ? fwd(i - 2 * NH) : i == 3 * NH ? P::kWFEATT : rev(i - 3 * NH - 1);
    }
    __host__ __device__ static constexpr int size(int i) {
        const int o = off(i);
        return o == P::kW0 ? a_block_floats(2, SDF_IN_STEPS) : o == P::kW0T ? a_block_floats(3, HS) : P::kHH;
    }
};

template <int NH, class Seq>
__device__ __forceinline__ void hidden_forward(float* stage, int op0, const float* __restrict__ wp, int lane, int h,
                                               const float (&in)[SDF_IN_STEPS], float (&sg)[NH][HS], float (&hlast)[HS],
                                               const Emitter* em = nullptr) {
    using P = SdfPack<NH>;
    f32x16 acc[2];
    load_vec<2>(wp + P::kB0, h, acc);
    gemm_staged<Seq, SDF_IN_STEPS, 2>(stage, wp, op0, lane, in, acc);
#pragma unroll
    for (int k = 1; k <= NH; ++k) {
        float d2;
        const float bound = acc_abs_max<2>(acc) + kSoftplusSlack;        // (form 2: the next GEMM's scale, known before its operands)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) softplus100_all(acc[t][r], hlast[16 * t + r], sg[k - 1][16 * t + r], d2);
        if (em) {
#pragma unroll
            for (int q = 0; q < HS; ++q) em->hid(SE<NH>::H(k), q, h, hlast[q]);
        }
        if (k < NH) {
            load_vec<2>(wp + P::bh(k), h, acc);
            gemm_staged<Seq, HS, 2>(stage, wp, op0 + k, lane, hlast, acc, &bound);
        }
    }
}

// reverse pass from the sdf output: fills dh[k-1] = dh_k for k = 1..NH-1 (dh_NH is the packed sdf row) and dl = dh_0.
template <int NH, class Seq>
__device__ __forceinline__ void reverse_pass(float* stage, int op0, const float* __restrict__ wp, int lane, int h,
                                             const float (&sg)[NH][HS], float (&dh)[NH > 1 ? NH - 1 : 1][HS], float (&dl)[48],
                                             const Emitter* em = nullptr) {
    using P = SdfPack<NH>;
    f32x16 ws[2];
    load_vec<2>(wp + P::kWSDF, h, ws);
    float da[HS];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) da[16 * t + r] = sg[NH - 1][16 * t + r] * ws[t][r];
    if (em) {
#pragma unroll
        for (int q = 0; q < HS; ++q) em->hid(SE<NH>::DA(NH), q, h, da[q]);
    }
    float bound = 0.0f;                                  // (form 2) scale hints: |da| <= |acc|, the Softplus derivative is a sigmoid
    const float* hint = nullptr;
#pragma unroll
    for (int k = NH - 1; k >= 1; --k) {
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        gemm_staged<Seq, HS, 2>(stage, wp, op0 + (NH - 1 - k), lane, da, acc, hint);
        bound = acc_abs_max<2>(acc);
        hint = &bound;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dh[k - 1][16 * t + r] = acc[t][r];
                da[16 * t + r] = sg[k - 1][16 * t + r] * acc[t][r];
            }
        if (em) {
#pragma unroll
            for (int q = 0; q < HS; ++q) em->hid(SE<NH>::DA(k), q, h, da[q]);
        }
    }
    f32x16 a3[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) a3[t][r] = 0.0f;
    gemm_staged<Seq, HS, 3>(stage, wp, op0 + NH - 1, lane, da, a3, hint);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dl[16 * t + r] = a3[t][r];
}

#ifndef NSA_OCC_FWD_FINE
#define NSA_OCC_FWD_FINE 2
#endif
#ifndef NSA_OCC_BWD_FINE
#define NSA_OCC_BWD_FINE 1
#endif
#ifndef NSA_OCC_BWD_COARSE
#define NSA_OCC_BWD_COARSE 2     // measured with LDS-staged weights: 59 us at two waves per SIMD vs 78 us at one
#endif
template <int L, int C, int NH>
__global__ __launch_bounds__(256, (NH == 1 ? 2 : NSA_OCC_FWD_FINE)) void k_sdfnet_fwd(SdfNetArgs a, GridGeom16 geom) {
    using P = SdfPack<NH>;
    using Seq = SdfOps<NH, false>;
    __shared__ __attribute__((aligned(16))) float stage[2 * kStageFloats];
    stage_begin<Seq>(stage, a.wp);
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    uint32_t tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t n_tiles = (a.src.P + 31) / 32;
    const bool wave_live = tile < n_tiles;                  // a wave without points still takes part in the barriers
    if (!wave_live) tile = n_tiles - 1;
    uint32_t pid = tile * 32 + (lane & 31);
    const bool live = wave_live && pid < a.src.P;
    if (pid >= a.src.P) pid = a.src.P - 1;
    const uint32_t q = point_of(a.src, pid);                // point handled by this lane pair
    float x[3], z;
    uint32_t ray;
    load_point(a.src, q, x, ray, z);

    float in[SDF_IN_STEPS];
    sdf_net_inputs<L, C>(x, a.divide_factor, a.table, geom, h, in);
    float sg[NH][HS], hl[HS];
    hidden_forward<NH, Seq>(stage, 0, a.wp, lane, h, in, sg, hl);
    // outputs: sdf (row 0, VALU dot) and the 64 features (rows 1..64)
    f32x16 ws[2], fo[2];
    load_vec<2>(a.wp + P::kWSDF, h, ws);
    float part = 0.0f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) part = fmaf(hl[16 * t + r], ws[t][r], part);
    float sdf = xhalf_sum(part) + a.wp[P::kBSDF];
    load_vec<2>(a.wp + P::kBFEAT, h, fo);
    gemm_staged<Seq, HS, 2>(stage, a.wp, NH, lane, hl, fo);
    float* fdst = a.feat + (size_t)tile * 32 * 64 + lane;
    if (wave_live) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = fo[t][r];
                if (a.accumulate) v += fdst[(16 * t + r) * 64];
                fdst[(16 * t + r) * 64] = v;
            }
    }
    // grad sdf
    float dh[NH > 1 ? NH - 1 : 1][HS], dl[48], g[3];
    reverse_pass<NH, Seq>(stage, NH + 1, a.wp, lane, h, sg, dh, dl);
    slots_to_x<L, C>(x, a.divide_factor, a.table, geom, h, in, dl, g);
#pragma unroll
    for (int d = 0; d < 3; ++d) g[d] = xhalf_sum(g[d]);
    if (live && h == 0) {
        if (a.accumulate) {
            sdf += a.sdf[q];
#pragma unroll
            for (int d = 0; d < 3; ++d) g[d] += a.grad[(size_t)q * 3 + d];
        }
        a.sdf[q] = sdf;
#pragma unroll
        for (int d = 0; d < 3; ++d) a.grad[(size_t)q * 3 + d] = g[d];
    }
}

// Occupancy per variant was chosen by A/B timing on MI355X (profiles/): forward 2 waves/SIMD, backward coarse 2, fine 1.
// MAP = true adds the mapping outputs: table gradient (run-merged atomics) and, for the coarse network, the per-point
// vectors of the weight-gradient GEMMs (the fine MLP is frozen in the reference, volsdf_train.py:150-173).
template <int L, int C, int NH, bool MAP>
// (the MAP variant keeps more state live -- emission, scatter -- and measured 20 % slower when capped at 256 VGPRs)
__global__ __launch_bounds__(256, (NH == 1 ? (MAP ? 1 : NSA_OCC_BWD_COARSE) : NSA_OCC_BWD_FINE)) void k_sdfnet_bwd(SdfNetArgs a, GridGeom16 geom) {
    using P = SdfPack<NH>;
    using Seq = SdfOps<NH, true>;
    __shared__ __attribute__((aligned(16))) float stage[2 * kStageFloats];
#define NSA_BODY_NW 4
#include "sdfnet_bwd_body.inc"
#undef NSA_BODY_NW
}

#ifndef NSA_SDFNET_AS_HEADER   // (render_colour.hip includes this file for the kernel pieces only: k_colour_coarse_bwd)
static int launch_sdfnet(bool bwd, const nsa_grid_t* grid, const SdfNetArgs& a, hipStream_t st) {
    const bool map = a.g_table != nullptr || a.emit != nullptr;
    GridGeom16 geom;
    if (int rc = make_grid_geom16(grid->offsets_host, grid->L, grid->S, grid->H, &geom, grid->C)) return rc;
    const uint32_t tiles = (a.src.P + 31) / 32;
    const dim3 g((tiles + 3) / 4), b(256);
    launch_begin();
    if (grid->L == 4 && grid->C == 8 && grid->n_hidden == 1) {
        if (bwd && map) hipLaunchKernelGGL((k_sdfnet_bwd<4, 8, 1, true>), g, b, 0, st, a, geom);
        else if (bwd)   hipLaunchKernelGGL((k_sdfnet_bwd<4, 8, 1, false>), g, b, 0, st, a, geom);
        else     hipLaunchKernelGGL((k_sdfnet_fwd<4, 8, 1>), g, b, 0, st, a, geom);
    } else if (grid->L == 8 && grid->C == 4 && grid->n_hidden == 3) {
        if (bwd && map) hipLaunchKernelGGL((k_sdfnet_bwd<8, 4, 3, true>), g, b, 0, st, a, geom);
        else if (bwd)   hipLaunchKernelGGL((k_sdfnet_bwd<8, 4, 3, false>), g, b, 0, st, a, geom);
        else     hipLaunchKernelGGL((k_sdfnet_fwd<8, 4, 3>), g, b, 0, st, a, geom);
    } else {
        return NSA_EUNSUPPORTED_NET;
    }
    return launch_end();
}
#endif  // NSA_SDFNET_AS_HEADER

}  // namespace nsa

#ifndef NSA_SDFNET_AS_HEADER
// Entry-point naming: this file is compiled twice -- as is (fp32-faithful GEMMs) and through *_bf16.hip with
// NSA_PIECES = 1, `nsa` renamed and every entry point suffixed _bf16; the fp32 entry points forward to those when
// nsa_grid_t.precision == 1.
#ifndef NSA_ENTRY
#define NSA_ENTRY(x) x
#endif
#include "bf16_entries.hpp"
#include "quad_entries.hpp"

extern "C" {

int NSA_ENTRY(nsa_sdfnet_forward)(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, int accumulate, float* sdf,
                       float* grad, float* feat_hl, nsa_stream_t stream) {
#if NSA_PIECES != 1
    if (grid && grid->precision == 1) return nsa_sdfnet_forward_bf16(pts, grid, packed, accumulate, sdf, grad, feat_hl, stream);      // bf16-operand kernels (csrc/*_bf16.hip)
#endif
    using namespace nsa;
    if (!pts || !grid || !packed || !sdf || !grad || !feat_hl) return NSA_EBADARG;
    if (pts->P == 0) return NSA_OK;
    if (!pts->points && (!pts->rays_o || !pts->rays_d || !pts->z_vals || pts->S == 0)) return NSA_EBADARG;
    if (grid->tile == 16) return NSA_ENTRY(nsa_sdfnet4_forward)(pts, grid, packed, accumulate, sdf, grad, feat_hl, stream);
    SdfNetArgs a{};
    a.src = PointSrc{pts->rays_o, pts->rays_d, pts->z_vals, pts->points, pts->P, pts->S, pts->order};
    a.table = grid->table; a.wp = packed; a.divide_factor = grid->divide_factor; a.accumulate = accumulate;
    a.sdf = sdf; a.grad = grad; a.feat = feat_hl;
    return launch_sdfnet(false, grid, a, (hipStream_t)stream);
}

int NSA_ENTRY(nsa_sdfnet_forward_pair)(const nsa_points_t* pts, const nsa_grid_t* coarse, const nsa_grid_t* fine,
                            const float* packed_coarse, const float* packed_fine, float* sdf, float* grad, float* feat_hl,
                            nsa_stream_t stream) {
#if NSA_PIECES != 1
    if (coarse && fine && coarse->precision == 1 && fine->precision == 1)
        return nsa_sdfnet_forward_pair_bf16(pts, coarse, fine, packed_coarse, packed_fine, sdf, grad, feat_hl, stream);
#endif
    using namespace nsa;
    if (!pts || !coarse || !fine || !packed_coarse || !packed_fine || !sdf || !grad || !feat_hl) return NSA_EBADARG;
    if (pts->P == 0) return NSA_OK;
    if (!pts->points && (!pts->rays_o || !pts->rays_d || !pts->z_vals || pts->S == 0)) return NSA_EBADARG;
    if (coarse->tile != 16 || fine->tile != 16 || coarse->precision != fine->precision) return NSA_EBADARG;   // quad packs, one precision
    return NSA_ENTRY(nsa_sdfnet4_forward_pair)(pts, coarse, fine, packed_coarse, packed_fine, sdf, grad, feat_hl, stream);
}

int NSA_ENTRY(nsa_sdfnet_backward)(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, const float* g_sdf,
                        const float* g_feat_hl, const float* g_grad, int accumulate, float* g_x, nsa_stream_t stream) {
#if NSA_PIECES != 1
    if (grid && grid->precision == 1) return nsa_sdfnet_backward_bf16(pts, grid, packed, g_sdf, g_feat_hl, g_grad, accumulate, g_x, stream);      // bf16-operand kernels (csrc/*_bf16.hip)
#endif
    using namespace nsa;
    if (!pts || !grid || !packed || !g_x) return NSA_EBADARG;
    if (pts->P == 0) return NSA_OK;
    if (!pts->points && (!pts->rays_o || !pts->rays_d || !pts->z_vals || pts->S == 0)) return NSA_EBADARG;
    if (grid->tile == 16)
        return NSA_ENTRY(nsa_sdfnet4_backward)(pts, grid, packed, g_sdf, g_feat_hl, g_grad, accumulate, g_x, nullptr, nullptr, 0, stream);
    SdfNetArgs a{};
    a.src = PointSrc{pts->rays_o, pts->rays_d, pts->z_vals, pts->points, pts->P, pts->S, pts->order};
    a.table = grid->table; a.wp = packed; a.divide_factor = grid->divide_factor; a.accumulate = accumulate;
    a.g_sdf = g_sdf; a.g_feat = g_feat_hl; a.g_grad = g_grad; a.g_x = g_x;
    return launch_sdfnet(true, grid, a, (hipStream_t)stream);
}

int NSA_ENTRY(nsa_sdfnet_backward_params)(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, const float* g_sdf,
                               const float* g_feat_hl, const float* g_grad, int accumulate, float* g_x, float* g_table,
                               float* emit, uint32_t emit_ld, nsa_stream_t stream) {
#if NSA_PIECES != 1
    if (grid && grid->precision == 1) return nsa_sdfnet_backward_params_bf16(pts, grid, packed, g_sdf, g_feat_hl, g_grad, accumulate, g_x, g_table, emit, emit_ld, stream);      // bf16-operand kernels (csrc/*_bf16.hip)
#endif
    using namespace nsa;
    if (!pts || !grid || !packed || !g_x || (!g_table && !emit)) return NSA_EBADARG;
    if (emit && emit_ld < ((pts->P + 31) / 32) * 32) return NSA_EBADARG;
    if (pts->P == 0) return NSA_OK;
    if (!pts->points && (!pts->rays_o || !pts->rays_d || !pts->z_vals || pts->S == 0)) return NSA_EBADARG;
    if (grid->tile == 16)
        return NSA_ENTRY(nsa_sdfnet4_backward)(pts, grid, packed, g_sdf, g_feat_hl, g_grad, accumulate, g_x, g_table, emit, emit_ld, stream);
    SdfNetArgs a{};
    a.src = PointSrc{pts->rays_o, pts->rays_d, pts->z_vals, pts->points, pts->P, pts->S, pts->order};
    a.table = grid->table; a.wp = packed; a.divide_factor = grid->divide_factor; a.accumulate = accumulate;
    a.g_sdf = g_sdf; a.g_feat = g_feat_hl; a.g_grad = g_grad; a.g_x = g_x;
    a.g_table = g_table; a.emit = emit; a.emit_ld = emit_ld;
    return launch_sdfnet(true, grid, a, (hipStream_t)stream);
}

int NSA_ENTRY(nsa_sdfnet_emit_rows)(void) { return nsa::SE<1>::ROWS; }
int NSA_ENTRY(nsa_sdfnet_emit_rows_nh)(uint32_t n_hidden) {
    return n_hidden == 1 ? nsa::SE<1>::ROWS : n_hidden == 3 ? nsa::SE<3>::ROWS : -1;
}
int NSA_ENTRY(nsa_sdfnet_emit_rows_tile)(uint32_t n_hidden, uint32_t tile) {
    return tile == 16 ? NSA_ENTRY(nsa_sdfnet4_emit_rows)(n_hidden) : NSA_ENTRY(nsa_sdfnet_emit_rows_nh)(n_hidden);
}

}  // extern "C"
#endif  // NSA_SDFNET_AS_HEADER
