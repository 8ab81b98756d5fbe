// render_colour.hip -- the colour network of the composite pass.
// Reference: RenderingNetwork.forward, mode "idr" with the colour hash grid (code/model/base_networks.py:333-395):
//   rgb = sigmoid(MLP_relu([x(3), PE4(view dir)(27), grad sdf(3), feature(64), colour-grid feature(32)]))   129 -> 64 -> 64 -> 3
// The colour grid is the 1 GiB, HBM-resident table (16 levels x 2 features, 7 hashed levels of 2^24 rows): each
// lane gathers 8 of the 16 levels (level 2*jl + h) with 8-byte loads, all 64 gathers of a lane in flight at once.
// The forward pass keeps what the backward needs from the table (the 16 features and 48 Jacobian entries of the
// lane's levels) in a per-lane save area, so the backward never touches the table again.
//
// Input-slot map (65 slots per half-wave h; reference column of rendering_input in brackets):
//   0..31          feature vector, HL order                                   [33 + f]
//   32             h0: x0 [0]        h1: x1 [1]
//   33             h0: x2 [2]        h1: d0 [3]
//   34             h0: d1 [4]        h1: d2 [5]
//   35             h0: grad0 [30]    h1: grad1 [31]
//   36             h0: grad2 [32]    h1: pad
//   37+2j, 38+2j   sin, cos of 2^k d_dd, pair g = 2j+h, k = g/3, dd = g%3    [6+6k+dd], [9+6k+dd]      (j < 6)
//   49+2jl+c       colour-grid level 2jl+h, channel c                         [97 + 2(2jl+h) + c]       (jl < 8)
#include <cstdlib>
#include "sdf_net.hpp"

namespace nsa {

constexpr int CL = 16, CC = 2;   // colour grid shape (hard-coded in the reference, base_networks.py:265-284)

struct ColPack {
    static constexpr int kHH = a_block_floats(2, HS);
    static constexpr int kW0 = 0;                                        // A[2][65 slots]
    static constexpr int kB0 = kW0 + a_block_floats(2, COL_IN_STEPS);
    static constexpr int kW1 = kB0 + 64;
    static constexpr int kB1 = kW1 + kHH;
    static constexpr int kW2V = kB1 + 64;                                // 3 output rows in activation layout
    static constexpr int kB2 = kW2V + 3 * 64;                            // [0..2]
    static constexpr int kW1T = kB2 + 64;
    static constexpr int kW0T = kW1T + kHH;                              // A[5][32]: rows = input slots
    static constexpr int kTotal = kW0T + a_block_floats(5, HS);
};

// Staged GEMM parts (mlp_common.hpp): the first layer (54 KiB packed) and its transpose (60 KiB) are split along k so that
// two 30 KiB stage buffers -- two workgroups per CU -- suffice.
//   forward : W0[g0-4], W0[g5-8], W1          backward: the same three (recompute), then W1^T, W0^T[g0-1], W0^T[g2-3]
constexpr int kColStage = 7680;     // floats per buffer: 2 tiles x 5 groups (= 5 tiles x 2 groups) x 3 KiB
template <bool BWD>
struct ColOps {
    static constexpr int n = BWD ? 6 : 3;
    __host__ __device__ static constexpr StageOp op(int i) {
        return i == 0 ? StageOp{ColPack::kW0, 2, 9, 0, 5} : i == 1 ? StageOp{ColPack::kW0, 2, 9, 5, 4}
             : i == 2 ? StageOp{ColPack::kW1, 2, 4, 0, 4} : i == 3 ? StageOp{ColPack::kW1T, 2, 4, 0, 4}
             : i == 4 ? StageOp{ColPack::kW0T, 5, 4, 0, 2} : StageOp{ColPack::kW0T, 5, 4, 2, 2};
    }
};

// Staged parts of a backward launch: the mapping form (MAP) recomputes the forward (all six parts); the data-path form needs the
// reverse GEMMs only (W1^T, W0^T in two parts) -- round 5: ReLU masks and the sigmoid outputs come from the forward's save area.
template <bool MAP>
struct ColBwdOps {
    static constexpr int n = MAP ? 6 : 3;
    static constexpr int kRev = MAP ? 3 : 0;                  // index of the W1^T part
    __host__ __device__ static constexpr StageOp op(int i) { return ColOps<true>::op(MAP ? i : i + 3); }
};

// Extension of the save area behind its ceil(P/32) x 4096 floats (features + Jacobian): per 32-point tile 256 floats --
//   [0..63] ReLU mask of layer 1 (bit 16 t + r of lane l = a1[t][r] > 0), [64..127] the same for layer 2, [128 + 32 j + p] rgb_j of
//   point p (the sigmoid outputs).  The data-path backward reads these instead of recomputing the forward MLP: it never needs the
//   first-layer input vector again (torch: relu'(a) = a > 0, sigmoid' = rgb (1 - rgb), base_networks.py:375-394).
constexpr int kSaveTile = 4096, kSaveExt = 256;
__device__ __forceinline__ float* save_ext(float* save, uint32_t P, uint32_t tile) {
    return save + (size_t)((P + 31) / 32) * kSaveTile + (size_t)tile * kSaveExt;
}

__device__ __forceinline__ uint32_t relu_mask(const f32x16 (&a)[2]) {
    uint32_t m = 0;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) m |= a[t][r] > 0.0f ? (1u << (16 * t + r)) : 0u;
    return m;
}

struct ColourArgs {
    PointSrc src;
    const float* table;
    const float* wp;
    float divide_factor;
    const float* grad;      // [P,3] grad sdf (input "normals")
    const float* feat;      // HL
    float* rgb;             // [P,3] out
    float* save;            // per-lane save area [tiles][64 floats][64 lanes] or nullptr (16 features + 48 Jacobian) + save_ext
    int save_no_features;   // 1: leave the 16 feature slots of the save area unwritten (only the mapping backward reads them, and the
                            //    tracker's forward entry is never followed by one): 16.8 MB less to write per 1024 x 128 batch
    // backward
    const float* g_rgb;     // [P,3]
    float* g_feat;          // HL out
    float* g_grad;          // [P,3] in/out: += colour part
    float* g_x;             // [P,3] out (overwrite)
    float* g_dir;           // [P,3] out (overwrite)
    int grid_grad;          // 0: colour-grid feature detached (color_stage "base"), 1: propagate through it
    // mapping (parameter gradients)
    float* g_table;         // colour-table gradient (atomically accumulated) or nullptr
    float* emit;            // per-point vectors for the weight-gradient GEMMs, [CE_ROWS][emit_ld] or nullptr
    uint32_t emit_ld;
};

// Rows of the emission buffer; column = tile*32 + point-in-tile.
//   dW0 = AB1 IN^T, db0 = sum AB1, dW1 = AB2 H1^T, db1 = sum AB2, dW2 = OB H2^T, db2 = sum OB
// IN rows are input slots (row = 2*slot + half), the others hidden features in reference order.
// (AB1, AB2, OB -- the regions whose row sums are the bias gradients -- are contiguous: one reduction)
enum : int { CE_IN = 0, CE_H1 = 130, CE_H2 = 194, CE_AB1 = 258, CE_AB2 = 322, CE_OB = 386, CE_ROWS = 389 };

struct ColEmitter {
    float* base;
    uint32_t ld;
    bool live;
    __device__ __forceinline__ void slot(int region, int s, int h, float v) const {
        base[(size_t)(region + 2 * s + h) * ld] = live ? v : 0.0f;
    }
    __device__ __forceinline__ void hid(int region, int q, int h, float v) const {
        const int f = 32 * (q >> 4) + (q & 3) + 8 * ((q & 15) >> 2) + 4 * h;
        base[(size_t)(region + f) * ld] = live ? v : 0.0f;
    }
};

// features of one colour-grid level into the first-layer slots and, with a save area, features + Jacobian for the backward
__device__ __forceinline__ void colour_level_out(const float (&v)[8][CC], const float (&w)[3], const float (&dw)[3], float scale,
                                                 bool inside, int jl, float (&in)[COL_IN_STEPS], float* sv, bool sv_features = true) {
    float f[CC];
    blend<3, CC>(v, w, f);
#pragma unroll
    for (int c = 0; c < CC; ++c) in[49 + jl * CC + c] = inside ? f[c] : 0.0f;
    if (sv) {
        if (sv_features) {
#pragma unroll
            for (int c = 0; c < CC; ++c) sv[(jl * CC + c) * 64] = in[49 + jl * CC + c];
        }
#pragma unroll
        for (int gd = 0; gd < 3; ++gd) {
            float jr[CC];
            jacobian_row<3, CC>(v, w, dw, scale, gd, jr);
#pragma unroll
            for (int c = 0; c < CC; ++c) sv[(16 + (jl * 3 + gd) * CC + c) * 64] = inside ? jr[c] : 0.0f;
        }
    }
}

// (An "x-pair" form that fetched the x / x+1 corner rows of the eight dense leading levels with one 16-byte load each was built in round 4
// and measured SLOWER -- 73.8 -> 83 us: the 16-byte loads sit at 8-byte alignment and the kernel needed 218 instead of 124 registers,
// two instead of four waves per SIMD (profiles/r04_ab_experiments.txt); removed in round 6.)
__device__ __forceinline__ void colour_inputs(const ColourArgs& a, const GridGeom16& geom, uint32_t tile, uint32_t q, int lane,
                                              int h, const float (&x)[3], const float (&dir)[3], float (&in)[COL_IN_STEPS],
                                              bool from_save, bool wave_live) {
    const float* fsrc = a.feat + (size_t)tile * 32 * 64 + lane;
#pragma unroll
    for (int q = 0; q < HS; ++q) in[q] = fsrc[q * 64];
    float gr[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) gr[d] = a.grad[(size_t)q * 3 + d];
    in[32] = h ? x[1] : x[0];
    in[33] = h ? dir[0] : x[2];
    in[34] = h ? dir[2] : dir[1];
    in[35] = h ? gr[1] : gr[0];
    in[36] = h ? 0.0f : gr[2];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int g0 = 2 * j, g1 = 2 * j + 1;
        const float da = h ? dir[g1 % 3] : dir[g0 % 3];
        const float sc = h ? (float)(1 << (g1 / 3)) : (float)(1 << (g0 / 3));
        sincos_f(da * sc, in[37 + 2 * j], in[38 + 2 * j]);
    }
    if (from_save) {
        const float* sv = a.save + (size_t)tile * 64 * 64 + lane;
#pragma unroll
        for (int q = 0; q < 16; ++q) in[49 + q] = sv[q * 64];
        return;
    }
    float u[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) u[d] = to_unit(x[d], a.divide_factor);
    float* sv = (a.save && wave_live) ? a.save + (size_t)tile * 64 * 64 + lane : nullptr;   // clamped waves write nothing
#pragma unroll
    for (int jl = 0; jl < CL / 2; ++jl) {
        const LevelGeom lg = geom.lv[2 * jl + h];
        uint32_t cell[3];
        float w[3], dw[3];
        const bool inside = locate<3>(u, lg.scale, cell, w, dw);
        float v[8][CC];
        gather_corners<3, CC>(a.table, lg, cell, v);
        colour_level_out(v, w, dw, lg.scale, inside, jl, in, sv, !a.save_no_features);
    }
}

// hidden activations: returns pre-activations a1, a2 (masks for the backward) and the 3 sigmoid outputs
// STAGED: weights through LDS (backward: 62 -> 56 us).  The forward kernel is bound by the HBM gather of the colour
// table and measured 3 us SLOWER with the staging barriers, so it streams its fragments from L2 per wave.
template <class Seq, bool STAGED>
__device__ __forceinline__ void colour_mlp(float* stage, const float* __restrict__ wp, int lane, int h,
                                           const float (&in)[COL_IN_STEPS], f32x16 (&a1)[2], f32x16 (&a2)[2], float (&rgb)[3],
                                           uint32_t* masks = nullptr) {
    load_vec<2>(wp + ColPack::kB0, h, a1);
    if (STAGED) {                                // (form 2: both parts of the first layer share one scale and one scaling of a1)
        const float m_in = abs_max<COL_IN_STEPS>(in);
        gemm_staged_part<Seq, kColStage, COL_IN_STEPS, 2, 0, 5, true, false>(stage, wp, 0, lane, in, a1, &m_in);
        gemm_staged_part<Seq, kColStage, COL_IN_STEPS, 2, 5, 4, false, true>(stage, wp, 1, lane, in, a1, &m_in);
    } else {
        gemm_op<COL_IN_STEPS, 2>(wp + ColPack::kW0, lane, in, a1);
    }
    const float m_h1 = acc_abs_max<2>(a1);       // (form 2 scale hint: a ReLU is at most |a|)
    float h1[HS];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) h1[16 * t + r] = relu_f(a1[t][r]);
    if (masks) masks[0] = relu_mask(a1);         // (taken where a1 dies: the forward keeps no pre-activation alive for it)
    load_vec<2>(wp + ColPack::kB1, h, a2);
    if (STAGED) gemm_staged_part<Seq, kColStage, HS, 2, 0, 4>(stage, wp, 2, lane, h1, a2, &m_h1);
    else        gemm_op<HS, 2>(wp + ColPack::kW1, lane, h1, a2, &m_h1);
    if (masks) masks[1] = relu_mask(a2);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        f32x16 wv[2];
        load_vec<2>(wp + ColPack::kW2V + 64 * j, h, wv);
        float part = 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) part = fmaf(relu_f(a2[t][r]), wv[t][r], part);
        const float o = xhalf_sum(part) + wp[ColPack::kB2 + j];
        rgb[j] = 1.0f / (1.0f + expf(-o));
    }
}

// The colour forward lives on memory-level parallelism (1 GiB table, HBM gather): four waves per SIMD.  Under a two-wave target the
// bf16-operand build allocated 133 registers and, with the ReLU masks of round 5, the fp32 build 134: a wave per SIMD lost.
#ifndef NSA_OCC_COL_FWD
#define NSA_OCC_COL_FWD 4
#endif
__global__ __launch_bounds__(256, NSA_OCC_COL_FWD) void k_colour_fwd(ColourArgs a, GridGeom16 geom) {
    using Seq = ColOps<false>;
#include "colour_fwd_body.inc"
}

#ifdef NSA_X_TS      // profiling build only (tools/ts_profile.py --colour)
static __device__ unsigned long long* g_ts_c = nullptr;
#define CTS_BEGIN const unsigned long long ts_start = ts_now(); unsigned long long ts_prev = ts_start; \
    if ((threadIdx.x & 63) == 0) for (int i = 0; i < 16; ++i) nsa_ts_lds[threadIdx.x >> 6][i] = 0;
#define CTS_MARK(slot) { const unsigned long long t_ = ts_now(); ts_add(slot, t_ - ts_prev); ts_prev = t_; }
#define CTS_END { ts_add(15, ts_now() - ts_start); if (g_ts_c && (threadIdx.x & 63) == 0) { \
    unsigned long long* o_ = g_ts_c + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16; \
    for (int i = 0; i < 16; ++i) o_[i] = nsa_ts_lds[threadIdx.x >> 6][i]; } }
#else
#define CTS_BEGIN
#define CTS_MARK(slot)
#define CTS_END
#endif
template <bool MAP>
__global__ __launch_bounds__(256, 2) void k_colour_bwd(ColourArgs a, GridGeom16 geom) {
    using Seq = ColBwdOps<MAP>;
    __shared__ __attribute__((aligned(16))) float stage[2 * kColStage];
    CTS_BEGIN
#define NSA_BODY_NW 4
#include "colour_bwd_body.inc"
#undef NSA_BODY_NW
    CTS_MARK(11)
    CTS_END
}

}  // namespace nsa

// ---- colour backward + coarse SDF backward as two phases of ONE 32-point launch (tracking: no parameter gradients) ------------------
// The two kernels tile the same points the same way (workgroup = 4 waves x 32 points), the coarse backward consumes what the colour
// backward has just written (feature and normal cotangents, d/dx to accumulate onto) and both run only two rounds of waves at the
// tracking batch: as one launch the input burst of a round is paid once, the coarse phase re-reads its cotangents from L2 and one
// launch disappears.  The bodies are the two kernels' own statements (colour_bwd_body.inc, sdfnet_bwd_body.inc): identical results.
#define NSA_SDFNET_AS_HEADER
#include "render_sdfnet.hip"
#undef NSA_SDFNET_AS_HEADER

namespace nsa {

// (4-wave workgroups, two per CU, as the two kernels run on their own: with one 8-wave workgroup per CU -- twice the waves behind every
//  staging barrier -- the launch measured 99.1 -> 106.7 us, profiles/r04_ab_experiments.txt r4t)
#define NSA_CC_NW 4
#define NSA_BODY_NW NSA_CC_NW
__global__ __launch_bounds__(64 * NSA_CC_NW, 2) void k_colour_coarse_bwd(ColourArgs ca, GridGeom16 cgeom, SdfNetArgs sa, GridGeom16 sgeom) {
    __shared__ __attribute__((aligned(16))) float stage[2 * kStageFloats];
    static_assert(kStageFloats >= kColStage, "the colour phase stages its parts in the SDF kernel's buffers");
    {   // phase 1: k_colour_bwd<false>
        constexpr bool MAP = false;
        using Seq = ColBwdOps<false>;
        const ColourArgs& a = ca;
        const GridGeom16& geom = cgeom;
#include "colour_bwd_body.inc"
    }
    __syncthreads();     // every wave is done with the colour phase's last staged part before the buffers are refilled; the feature /
                         // normal cotangents and d/dx this workgroup wrote are complete (vmcnt(0) + barrier)
    {   // phase 2: k_sdfnet_bwd<4, 8, 1, false> on the same tiles
        constexpr int L = 4, C = 8, NH = 1;
        constexpr bool MAP = false;
        using P = SdfPack<NH>;
        using Seq = SdfOps<NH, true>;
        const SdfNetArgs& a = sa;
        const GridGeom16& geom = sgeom;
#include "sdfnet_bwd_body.inc"
    }
}
#undef NSA_BODY_NW

}  // namespace nsa

// ---- colour forward + the ray's composite / L1 / composite backward as two phases of ONE launch (tracker, 128 samples per ray) -------
// A workgroup of the colour forward is 4 waves x 32 points = the 128 samples of ONE ray (ray order, P a multiple of 128): when its colours
// are complete, its first wave runs k_composite_track for that ray -- the launch-bound per-ray kernel (7.5 us on an idle GPU) disappears
// into the tail of the one-round colour forward.  The two kernels' own statements (colour_fwd_body.inc, composite_track_body.inc).
#define NSA_COMPOSITE_AS_HEADER
#include "render_composite.hip"
#undef NSA_COMPOSITE_AS_HEADER

namespace nsa {

__global__ __launch_bounds__(256, NSA_OCC_COL_FWD) void k_colour_fwd_track(ColourArgs ca, GridGeom16 cgeom, CompositeArgs ta,
                                                                            const float* __restrict__ gt, float* __restrict__ ray_loss,
                                                                            float inv_n) {
    {   // phase 1: k_colour_fwd (every wave is live: the entry point requires P % 128 == 0)
        using Seq = ColOps<false>;
        const ColourArgs& a = ca;
        const GridGeom16& geom = cgeom;
#include "colour_fwd_body.inc"
    }
    __syncthreads();     // the ray's 128 colours are stored (vmcnt(0) + barrier)
    if ((threadIdx.x >> 6) == 0) {   // phase 2: k_composite_track for ray blockIdx.x
        const CompositeArgs& a = ta;
        const int lane = threadIdx.x & 63;
        const uint32_t ray = blockIdx.x;
#include "composite_track_body.inc"
    }
}

// the same with the generic composite forward (all five ray outputs + the weights) as the second phase: the autograd path
__global__ __launch_bounds__(256, NSA_OCC_COL_FWD) void k_colour_fwd_composite(ColourArgs ca, GridGeom16 cgeom, CompositeArgs ta) {
    {   // phase 1: k_colour_fwd (every wave is live: the entry point requires P % 128 == 0)
        using Seq = ColOps<false>;
        const ColourArgs& a = ca;
        const GridGeom16& geom = cgeom;
#include "colour_fwd_body.inc"
    }
    __syncthreads();
    if ((threadIdx.x >> 6) == 0) {   // phase 2: k_composite_fwd for ray blockIdx.x
        const CompositeArgs& a = ta;
        const int lane = threadIdx.x & 63;
        const uint32_t ray = blockIdx.x;
#include "composite_fwd_body.inc"
    }
}

}  // namespace nsa

#if defined(NSA_X_TS) && NSA_PIECES != 1
extern "C" int nsa_debug_set_ts_colour(unsigned long long* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(nsa::g_ts_c), &p, sizeof(p)) == hipSuccess ? 0 : 3;
}
#endif

// Entry-point naming: this file is compiled twice -- as is (fp32-faithful GEMMs) and through *_bf16.hip with
// NSA_PIECES = 1, `nsa` renamed and every entry point suffixed _bf16; the fp32 entry points forward to those when
// nsa_grid_t.precision == 1.
#ifndef NSA_ENTRY
#define NSA_ENTRY(x) x
#endif
#include "bf16_entries.hpp"

extern "C" {

static int colour_common(const nsa_points_t* pts, const nsa_grid_t* grid, nsa::ColourArgs* a, nsa::GridGeom16* geom) {
    using namespace nsa;
    if (!pts || !grid) return NSA_EBADARG;
    if (pts->points || !pts->rays_o || !pts->rays_d || !pts->z_vals || pts->S == 0) return NSA_EBADARG;   // needs view dirs
    if (!(grid->L == 16 && grid->C == 2)) return NSA_EUNSUPPORTED_NET;
    if (int rc = make_grid_geom16(grid->offsets_host, grid->L, grid->S, grid->H, geom, grid->C)) return rc;
    a->src = PointSrc{pts->rays_o, pts->rays_d, pts->z_vals, nullptr, pts->P, pts->S, pts->order};
    a->table = grid->table;
    a->divide_factor = grid->divide_factor;
    return NSA_OK;
}

int NSA_ENTRY(nsa_colour_forward)(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, const float* grad,
                       const float* feat_hl, float* rgb, float* save, nsa_stream_t stream) {
#if NSA_PIECES != 1
    if (grid && grid->precision == 1) return nsa_colour_forward_bf16(pts, grid, packed, grad, feat_hl, rgb, save, stream);      // bf16-operand kernels (csrc/*_bf16.hip)
#endif
    using namespace nsa;
    if (!packed || !grad || !feat_hl || !rgb) return NSA_EBADARG;
    ColourArgs a{};
    GridGeom16 geom;
    if (int rc = colour_common(pts, grid, &a, &geom)) return rc;
    if (pts->P == 0) return NSA_OK;
    a.wp = packed; a.grad = grad; a.feat = feat_hl; a.rgb = rgb; a.save = save;
    const uint32_t tiles = (pts->P + 31) / 32;
    launch_begin();
    hipLaunchKernelGGL(k_colour_fwd, dim3((tiles + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, geom);
    return launch_end();
}

int NSA_ENTRY(nsa_colour_forward_composite)(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, const float* grad,
                                 const float* feat_hl, float* rgb, float* save, const float* sdf, const float* voxels,
                                 uint32_t voxel_res, float* weights, float* rgb_values, float* depth, float* nmap, float* entropy,
                                 nsa_stream_t stream) {
#if NSA_PIECES != 1
    if (grid && grid->precision == 1)
        return nsa_colour_forward_composite_bf16(pts, grid, packed, grad, feat_hl, rgb, save, sdf, voxels, voxel_res, weights, rgb_values,
                                                 depth, nmap, entropy, stream);
#endif
    using namespace nsa;
    if (!packed || !grad || !feat_hl || !rgb || !sdf || !voxels || !weights || !rgb_values || !depth || !nmap || !entropy)
        return NSA_EBADARG;
    ColourArgs a{};
    GridGeom16 geom;
    if (int rc = colour_common(pts, grid, &a, &geom)) return rc;
    if (pts->P == 0) return NSA_OK;
    if (pts->points || pts->order || pts->S != 128 || pts->P % 128 != 0) return NSA_EBADARG;      // one workgroup = one ray
    a.wp = packed; a.grad = grad; a.feat = feat_hl; a.rgb = rgb; a.save = save;
    CompositeArgs t{};
    t.rays_o = pts->rays_o; t.rays_d = pts->rays_d; t.z_vals = pts->z_vals; t.sdf = sdf; t.rgb = rgb; t.grad = grad; t.voxels = voxels;
    t.voxel_res = voxel_res; t.R = pts->P / 128; t.S = 128;
    t.weights = weights; t.rgb_values = rgb_values; t.depth = depth; t.nmap = nmap; t.entropy = entropy;
    launch_begin();
    hipLaunchKernelGGL(k_colour_fwd_composite, dim3(pts->P / 128), dim3(256), 0, (hipStream_t)stream, a, geom, t);
    return launch_end();
}

int NSA_ENTRY(nsa_colour_forward_track)(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, const float* grad,
                             const float* feat_hl, float* rgb, float* save, const float* sdf, const float* voxels, uint32_t voxel_res,
                             const float* gt, uint32_t n_total, float* rgb_values, float* ray_loss, float* g_sdf, float* g_rgb,
                             float* g_grad, nsa_stream_t stream) {
#if NSA_PIECES != 1
    if (grid && grid->precision == 1)
        return nsa_colour_forward_track_bf16(pts, grid, packed, grad, feat_hl, rgb, save, sdf, voxels, voxel_res, gt, n_total, rgb_values,
                                             ray_loss, g_sdf, g_rgb, g_grad, stream);
#endif
    using namespace nsa;
    if (!packed || !grad || !feat_hl || !rgb || !sdf || !voxels || !gt || !rgb_values || !ray_loss || !g_sdf || !g_rgb || !g_grad)
        return NSA_EBADARG;
    ColourArgs a{};
    GridGeom16 geom;
    if (int rc = colour_common(pts, grid, &a, &geom)) return rc;
    if (pts->P == 0) return NSA_OK;
    // one workgroup = one ray: ray samples in ray order, 128 per ray
    if (pts->points || pts->order || pts->S != 128 || pts->P % 128 != 0 || n_total < pts->P / 128) return NSA_EBADARG;
    a.wp = packed; a.grad = grad; a.feat = feat_hl; a.rgb = rgb; a.save = save;
#ifndef NSA_X_SAVE_FEATURES      // (experiment builds: A/B of the feature stores)
    a.save_no_features = 1;      // this entry is the tracker's: its backward is the data-path one (ReLU masks, outputs, Jacobian)
#endif
    CompositeArgs t{};
    t.rays_o = pts->rays_o; t.rays_d = pts->rays_d; t.z_vals = pts->z_vals; t.sdf = sdf; t.rgb = rgb; t.voxels = voxels;
    t.voxel_res = voxel_res; t.R = pts->P / 128; t.S = 128;
    t.rgb_values = rgb_values; t.g_sdf = g_sdf; t.g_rgb = g_rgb; t.g_grad = g_grad;
    launch_begin();
    hipLaunchKernelGGL(k_colour_fwd_track, dim3(pts->P / 128), dim3(256), 0, (hipStream_t)stream, a, geom, t, gt, ray_loss,
                       1.0f / (float)(3 * (uint64_t)n_total));
    return launch_end();
}

int NSA_ENTRY(nsa_colour_backward)(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, const float* grad,
                        const float* feat_hl, const float* save, const float* g_rgb, int grid_grad, float* g_feat_hl,
                        float* g_grad, float* g_x, float* g_dir, nsa_stream_t stream) {
#if NSA_PIECES != 1
    if (grid && grid->precision == 1) return nsa_colour_backward_bf16(pts, grid, packed, grad, feat_hl, save, g_rgb, grid_grad, g_feat_hl, g_grad, g_x, g_dir, stream);      // bf16-operand kernels (csrc/*_bf16.hip)
#endif
    using namespace nsa;
    if (!packed || !grad || !feat_hl || !save || !g_rgb || !g_feat_hl || !g_grad || !g_x || !g_dir) return NSA_EBADARG;
    ColourArgs a{};
    GridGeom16 geom;
    if (int rc = colour_common(pts, grid, &a, &geom)) return rc;
    if (pts->P == 0) return NSA_OK;
    a.wp = packed; a.grad = grad; a.feat = feat_hl; a.save = const_cast<float*>(save); a.g_rgb = g_rgb;
    a.g_feat = g_feat_hl; a.g_grad = g_grad; a.g_x = g_x; a.g_dir = g_dir; a.grid_grad = grid_grad;
    const uint32_t tiles = (pts->P + 31) / 32;
    launch_begin();
    hipLaunchKernelGGL(k_colour_bwd<false>, dim3((tiles + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, geom);
    return launch_end();
}

int NSA_ENTRY(nsa_colour_coarse_backward)(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, const float* grad,
                               const float* feat_hl, const float* save, const float* g_rgb, int grid_grad, float* g_feat_hl,
                               float* g_grad, float* g_x, float* g_dir, const nsa_grid_t* coarse, const float* packed_coarse,
                               const float* g_sdf, nsa_stream_t stream) {
#if NSA_PIECES != 1
    if (grid && coarse && grid->precision == 1 && coarse->precision == 1)
        return nsa_colour_coarse_backward_bf16(pts, grid, packed, grad, feat_hl, save, g_rgb, grid_grad, g_feat_hl, g_grad, g_x, g_dir, coarse,
                                               packed_coarse, g_sdf, stream);
#endif
    using namespace nsa;
    if (!packed || !grad || !feat_hl || !save || !g_rgb || !g_feat_hl || !g_grad || !g_x || !g_dir || !coarse || !packed_coarse)
        return NSA_EBADARG;
    if (grid->precision != coarse->precision) return NSA_EBADARG;
    // the coarse network in the 32-point tiling (the tiles of the colour kernels), 4 levels x 8 channels, one hidden layer
    if (coarse->tile == 16 || !(coarse->L == 4 && coarse->C == 8 && coarse->n_hidden == 1)) return NSA_EUNSUPPORTED_NET;
    ColourArgs a{};
    GridGeom16 geom, sgeom;
    if (int rc = colour_common(pts, grid, &a, &geom)) return rc;
    if (pts->P == 0) return NSA_OK;
    if (int rc = make_grid_geom16(coarse->offsets_host, coarse->L, coarse->S, coarse->H, &sgeom, coarse->C)) return rc;
    a.wp = packed; a.grad = grad; a.feat = feat_hl; a.save = const_cast<float*>(save); a.g_rgb = g_rgb;
    a.g_feat = g_feat_hl; a.g_grad = g_grad; a.g_x = g_x; a.g_dir = g_dir; a.grid_grad = grid_grad;
    SdfNetArgs sa{};
    sa.src = a.src;
    sa.table = coarse->table; sa.wp = packed_coarse; sa.divide_factor = coarse->divide_factor; sa.accumulate = 1;
    sa.g_sdf = g_sdf; sa.g_feat = g_feat_hl; sa.g_grad = g_grad; sa.g_x = g_x;
    const uint32_t tiles = (pts->P + 31) / 32;
    launch_begin();
    hipLaunchKernelGGL(k_colour_coarse_bwd, dim3((tiles + NSA_CC_NW - 1) / NSA_CC_NW), dim3(64 * NSA_CC_NW), 0, (hipStream_t)stream, a, geom, sa, sgeom);
    return launch_end();
}

int NSA_ENTRY(nsa_colour_backward_params)(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, const float* grad,
                               const float* feat_hl, const float* save, const float* g_rgb, int grid_grad,
                               float* g_feat_hl, float* g_grad, float* g_x, float* g_dir, float* g_table, float* emit,
                               uint32_t emit_ld, nsa_stream_t stream) {
#if NSA_PIECES != 1
    if (grid && grid->precision == 1) return nsa_colour_backward_params_bf16(pts, grid, packed, grad, feat_hl, save, g_rgb, grid_grad, g_feat_hl, g_grad, g_x, g_dir, g_table, emit, emit_ld, stream);      // bf16-operand kernels (csrc/*_bf16.hip)
#endif
    using namespace nsa;
    if (!packed || !grad || !feat_hl || !save || !g_rgb || !g_feat_hl || !g_grad || !g_x || !g_dir) return NSA_EBADARG;
    if (!g_table && !emit) return NSA_EBADARG;
    ColourArgs a{};
    GridGeom16 geom;
    if (int rc = colour_common(pts, grid, &a, &geom)) return rc;
    if (pts->P == 0) return NSA_OK;
    if (emit && emit_ld < ((pts->P + 31) / 32) * 32) return NSA_EBADARG;
    a.wp = packed; a.grad = grad; a.feat = feat_hl; a.save = const_cast<float*>(save); a.g_rgb = g_rgb;
    a.g_feat = g_feat_hl; a.g_grad = g_grad; a.g_x = g_x; a.g_dir = g_dir; a.grid_grad = grid_grad;
    a.g_table = g_table; a.emit = emit; a.emit_ld = emit_ld;
    const uint32_t tiles = (pts->P + 31) / 32;
    launch_begin();
    hipLaunchKernelGGL(k_colour_bwd<true>, dim3((tiles + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, geom);
    return launch_end();
}

int NSA_ENTRY(nsa_colour_emit_rows)(void) { return nsa::CE_ROWS; }

}  // extern "C"
