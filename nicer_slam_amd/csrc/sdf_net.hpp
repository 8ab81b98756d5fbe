// sdf_net.hpp -- the SDF networks (coarse: 71->64->65, fine: 71->64->64->64->65; Softplus(100)) evaluated per
// wave on MFMA, fused with their grid encoders and positional encoding.
// Reference: ImplicitNetworkGrid.forward / get_outputs (code/model/base_networks.py:155-221), HashEncoder.forward
// (code/hashencoder/hashgrid.py:199-215), Embedder (code/model/embedder.py:5-37).
//
// Input-slot map of the first layer (36 slots per half-wave h; reference feature index in brackets):
//   slot 0        : h=0 -> x0 [0]              h=1 -> x2 [2]
//   slot 1        : h=0 -> x1 [1]              h=1 -> zero pad
//   slot 2+2j,3+2j: sin, cos of 2^k x_d for the pair g = 2j+h, k = g/3, d = g%3   [3+6k+d], [6+6k+d]     (j < 9)
//   slot 20+jl*C+c: channel c of grid level 2*jl+h   [39 + (2jl+h)*C + c]                               (jl < L/2)
// (nicer_slam_amd/fused/pack.py builds the packed weights from exactly this table.)
#pragma once
#include "grid_common.hpp"
#include "mlp_common.hpp"

namespace nsa {

struct GridGeom16 {
    LevelGeom lv[16];
};

inline int make_grid_geom16(const int32_t* offsets_host, uint32_t L, float S, uint32_t H, GridGeom16* out, uint32_t C) {
    if (L > 16) return NSA_ETOO_MANY_LEVELS;
    GridGeom g;
    if (int rc = make_grid_geom(offsets_host, L, 3, S, H, &g, C)) return rc;
    for (uint32_t l = 0; l < L; ++l) out->lv[l] = g.lv[l];
    if (has_generic_level(out->lv, L)) return NSA_EUNSUPPORTED_NET;   // fused kernels carry no generic-modulo path
    return NSA_OK;
}

// Offsets (in floats) inside one SDF net's packed parameter block; NH = number of hidden layers.
template <int NH>
struct SdfPack {
    static constexpr int kHH = a_block_floats(2, HS);                // one 64x64 hidden block
    static constexpr int kW0 = 0;                                   // A[2 tiles][36 slots]
    static constexpr int kB0 = kW0 + a_block_floats(2, SDF_IN_STEPS);
    static constexpr int kWH = kB0 + 64;                            // (NH-1) x { A[2][32], bias[64] }
    static constexpr int kWSDF = kWH + (NH - 1) * (kHH + 64);       // last-layer row 0 in activation layout
    static constexpr int kBSDF = kWSDF + 64;                        // [0] = bias of the sdf output
    static constexpr int kWFEAT = kBSDF + 64;                       // last-layer rows 1..64: A[2][32]
    static constexpr int kBFEAT = kWFEAT + kHH;
    static constexpr int kWHT = kBFEAT + 64;                        // transposed hidden layers, order k = NH-1 .. 1
    static constexpr int kW0T = kWHT + (NH - 1) * kHH;              // A[3 tiles][32 slots]: rows = input slots
    static constexpr int kWFEATT = kW0T + a_block_floats(3, HS);    // transposed feature rows
    static constexpr int kTotal = kWFEATT + kHH;
    __host__ __device__ static constexpr int wh(int k) { return kWH + (k - 1) * (kHH + 64); }       // k = 1..NH-1
    __host__ __device__ static constexpr int bh(int k) { return wh(k) + kHH; }
    __host__ __device__ static constexpr int wht(int k) { return kWHT + (NH - 1 - k) * kHH; }       // k = 1..NH-1
};

// Position + positional-encoding slots 0..19 (identical for the coarse and the fine network).
__device__ __forceinline__ void pe_slots(const float (&x)[3], int h, float (&in)[SDF_IN_STEPS]) {
    in[0] = h ? x[2] : x[0];
    in[1] = h ? 0.0f : x[1];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int g0 = 2 * j, g1 = 2 * j + 1;
        const float xa = h ? x[g1 % 3] : x[g0 % 3];
        const float sc = h ? (float)(1 << (g1 / 3)) : (float)(1 << (g0 / 3));
        sincos_f(xa * sc, in[2 + 2 * j], in[3 + 2 * j]);
    }
}

// Grid-feature slots 20..35: the L/2 levels (2*jl + h) of this lane, C channels each.
// jstore != nullptr: also keeps d feature / d u (the 3 x C Jacobian rows of every level of this lane, zero outside the
// grid) in a lane-private LDS column, jstore[((jl*3 + d)*C + c) * 64] -- the backward then needs no second and third
// corner gather (slots_to_x_jac / tangent_from_jac below).
template <int L, int C, bool FAST = false>
__device__ __forceinline__ void grid_slots(const float (&x)[3], float divide_factor, const float* __restrict__ table,
                                           const GridGeom16& geom, int h, float (&in)[SDF_IN_STEPS], float* jstore = nullptr) {
    float u[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) u[d] = to_unit(x[d], divide_factor);   // hashgrid.py:203 (size = 1)
#pragma unroll
    for (int jl = 0; jl < L / 2; ++jl) {
        const LevelGeom g = geom.lv[2 * jl + h];
        uint32_t cell[3];
        float w[3], dw[3];
        const bool inside = locate<3>(u, g.scale, cell, w, dw);
        float v[8][C];
        gather_corners<3, C, false, FAST>(table, g, cell, v);
        float f[C];
        blend<3, C>(v, w, f);
#pragma unroll
        for (int c = 0; c < C; ++c) in[20 + jl * C + c] = inside ? f[c] : 0.0f;
        if (jstore) {
#pragma unroll
            for (int gd = 0; gd < 3; ++gd) {
                float jr[C];
                jacobian_row<3, C>(v, w, dw, g.scale, gd, jr);
#pragma unroll
                for (int c = 0; c < C; ++c) jstore[((jl * 3 + gd) * C + c) * 64] = inside ? jr[c] : 0.0f;
            }
        }
    }
}

// First-layer inputs of one SDF net for one point, as seen by lane half h.
template <int L, int C>
__device__ __forceinline__ void sdf_net_inputs(const float (&x)[3], float divide_factor, const float* __restrict__ table,
                                               const GridGeom16& geom, int h, float (&in)[SDF_IN_STEPS],
                                               float* jstore = nullptr) {
    pe_slots(x, h, in);
    grid_slots<L, C>(x, divide_factor, table, geom, h, in, jstore);
}

// NOTE on out-of-range points: the table gather above still runs for them (addresses stay inside the level because
// of the modulo), only the result is zeroed -- same values as kernel_grid's early-out (hashencoder.cu:161-177).
// A point with NaN coordinates is "inside" for the reference's </> tests; cell then comes from (uint)NaN: keep the
// gather in range by construction of level_row (mask / modulo), values are garbage-in garbage-out on both sides.

// SDF value only (sampler pass): acc chain through the hidden layers, then the sdf row as a VALU dot.
template <int NH>
__device__ __forceinline__ float sdf_only(const float* __restrict__ wp, int lane, int h, const float (&in)[SDF_IN_STEPS]) {
    using P = SdfPack<NH>;
    f32x16 acc[2];
    load_vec<2>(wp + P::kB0, h, acc);
    gemm_op<SDF_IN_STEPS, 2>(wp + P::kW0, lane, in, acc);
    float act[HS];
#pragma unroll
    for (int k = 1; k < NH; ++k) {
        const float bound = acc_abs_max<2>(acc) + kSoftplusSlack;       // (form 2: the next GEMM's scale, known before its operands)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) act[16 * t + r] = softplus100(acc[t][r]);
        load_vec<2>(wp + P::bh(k), h, acc);
        gemm_op<HS, 2>(wp + P::wh(k), lane, act, acc, &bound);
    }
    f32x16 ws[2];
    load_vec<2>(wp + P::kWSDF, h, ws);
    float part = 0.0f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) part = fmaf(softplus100(acc[t][r]), ws[t][r], part);
    return xhalf_sum(part) + wp[P::kBSDF];
}

// The same for T point tiles per wave (gemm_op_tiles): in[t] / the result of tile t.
template <int NH, int T>
__device__ __forceinline__ void sdf_only_tiles(const float* __restrict__ wp, int lane, int h, const float (&in)[T][SDF_IN_STEPS],
                                               float (&sdf)[T]) {
    using P = SdfPack<NH>;
    f32x16 acc[T][2];
#pragma unroll
    for (int t = 0; t < T; ++t) load_vec<2>(wp + P::kB0, h, acc[t]);
    gemm_op_tiles<SDF_IN_STEPS, 2, T>(wp + P::kW0, lane, in, acc);
    float act[T][HS];
#pragma unroll
    for (int k = 1; k < NH; ++k) {
        float bound[T];                                                  // (form 2: the next GEMM's scales, known before its operands)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            bound[t] = acc_abs_max<2>(acc[t]) + kSoftplusSlack;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) act[t][16 * mt + r] = softplus100(acc[t][mt][r]);
            load_vec<2>(wp + P::bh(k), h, acc[t]);
        }
        gemm_op_tiles<HS, 2, T>(wp + P::wh(k), lane, act, acc, bound);
    }
    f32x16 ws[2];
    load_vec<2>(wp + P::kWSDF, h, ws);
    const float bias = wp[P::kBSDF];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        float part = 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) part = fmaf(softplus100(acc[t][mt][r]), ws[mt][r], part);
        sdf[t] = xhalf_sum(part) + bias;
    }
}

}  // namespace nsa

namespace nsa {

// ------------------------------------------------------------------------------------------------------------------
// Contraction of a per-slot cotangent with d(first-layer input)/dx for THIS lane's share of the slots:
//   g[d] = sum_slots dl[slot] * d in[slot] / d x_d     (x slots, sin/cos pairs, grid levels 2*jl+h through J/(2 df)).
// Used for grad sdf (dl = reverse pass from the sdf output, base_networks.py:214-219) and for the value-path input
// gradient (dl = d loss / d h0).  The grid Jacobian is not stored: the corner rows are gathered again (L2-resident
// SDF tables) and contracted as  sum_faces w_face * (p_hi - p_lo),  p_corner = <dl_level, row_corner>.
// Caller adds the two half-waves (xhalf_sum).
template <int L, int C>
__device__ __forceinline__ void slots_to_x(const float (&x)[3], float divide_factor, const float* __restrict__ table,
                                           const GridGeom16& geom, int h, const float (&in)[SDF_IN_STEPS],
                                           const float (&dl)[3 * 16], float (&g)[3]) {
    g[0] = h ? 0.0f : dl[0];
    g[1] = h ? 0.0f : dl[1];
    g[2] = h ? dl[0] : 0.0f;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        constexpr int unused = 0; (void)unused;
        const int g0 = 2 * j, g1 = 2 * j + 1;
        const float sc = h ? (float)(1 << (g1 / 3)) : (float)(1 << (g0 / 3));
        const float t = sc * (in[3 + 2 * j] * dl[2 + 2 * j] - in[2 + 2 * j] * dl[3 + 2 * j]);   // 2^k (cos d_sin - sin d_cos)
        g[g0 % 3] += h ? 0.0f : t;
        g[g1 % 3] += h ? t : 0.0f;
    }
    float u[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) u[d] = to_unit(x[d], divide_factor);
    const float chain = 1.0f / (2.0f * divide_factor);
#pragma unroll
    for (int jl = 0; jl < L / 2; ++jl) {
        const LevelGeom lg = geom.lv[2 * jl + h];
        uint32_t cell[3];
        float w[3], dw[3];
        const bool inside = locate<3>(u, lg.scale, cell, w, dw);
        float v[8][C];
        gather_corners<3, C>(table, lg, cell, v);
        float p[8];
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            p[corner] = 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) p[corner] = fmaf(dl[20 + jl * C + c], v[corner][c], p[corner]);
        }
#pragma unroll
        for (int gd = 0; gd < 3; ++gd) {
            float acc = 0.0f;
#pragma unroll
            for (int face = 0; face < 4; ++face) {
                float wt = lg.scale;
                int lo = 0;
#pragma unroll
                for (int nd = 0; nd < 2; ++nd) {
                    const int d = (nd >= gd) ? nd + 1 : nd;
                    if ((face >> nd) & 1) { wt *= w[d]; lo |= 1 << d; }
                    else                  { wt *= 1.0f - w[d]; }
                }
                acc = fmaf(wt, p[lo | (1 << gd)] - p[lo], acc);
            }
            g[gd] += inside ? acc * dw[gd] * chain : 0.0f;
        }
    }
}

// Tangent of the first-layer input along n (a cotangent of grad sdf): dn_in[slot] = sum_d (d in[slot]/d x_d) n_d.
// This is the reverse of the grad-sdf assembly w.r.t. the reverse-pass vector; the grid Hessian term is NOT part of
// it (hashgrid.py:134).  Also returns the positional-encoding second-derivative term
//   xbar_d += n_d * sum_k -(4^k) (sin * dl_sin + cos * dl_cos)      (dl = reverse-pass vector of this point).
template <int L, int C>
__device__ __forceinline__ void x_to_slots_tangent(const float (&x)[3], float divide_factor, const float* __restrict__ table,
                                                   const GridGeom16& geom, int h, const float (&in)[SDF_IN_STEPS],
                                                   const float (&n)[3], const float (&dl)[3 * 16],
                                                   float (&tin)[SDF_IN_STEPS], float (&xbar)[3]) {
    tin[0] = h ? n[2] : n[0];
    tin[1] = h ? 0.0f : n[1];
    xbar[0] = xbar[1] = xbar[2] = 0.0f;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int g0 = 2 * j, g1 = 2 * j + 1;
        const float sc = h ? (float)(1 << (g1 / 3)) : (float)(1 << (g0 / 3));
        const float nd = h ? n[g1 % 3] : n[g0 % 3];
        const float s = in[2 + 2 * j], c = in[3 + 2 * j];
        tin[2 + 2 * j] = sc * c * nd;
        tin[3 + 2 * j] = -sc * s * nd;
        const float t = -sc * sc * (s * dl[2 + 2 * j] + c * dl[3 + 2 * j]) * nd;
        xbar[g0 % 3] += h ? 0.0f : t;
        xbar[g1 % 3] += h ? t : 0.0f;
    }
    float u[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) u[d] = to_unit(x[d], divide_factor);
    const float chain = 1.0f / (2.0f * divide_factor);
#pragma unroll
    for (int jl = 0; jl < L / 2; ++jl) {
        const LevelGeom lg = geom.lv[2 * jl + h];
        uint32_t cell[3];
        float w[3], dw[3];
        const bool inside = locate<3>(u, lg.scale, cell, w, dw);
        float v[8][C];
        gather_corners<3, C>(table, lg, cell, v);
        float k[8];
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) k[corner] = 0.0f;
#pragma unroll
        for (int gd = 0; gd < 3; ++gd) {
#pragma unroll
            for (int face = 0; face < 4; ++face) {
                float wt = lg.scale;
                int lo = 0;
#pragma unroll
                for (int nd = 0; nd < 2; ++nd) {
                    const int d = (nd >= gd) ? nd + 1 : nd;
                    if ((face >> nd) & 1) { wt *= w[d]; lo |= 1 << d; }
                    else                  { wt *= 1.0f - w[d]; }
                }
                const float t = wt * dw[gd] * n[gd] * chain;
                k[lo | (1 << gd)] += t;
                k[lo] -= t;
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float acc = 0.0f;
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) acc = fmaf(k[corner], v[corner][c], acc);
            tin[20 + jl * C + c] = inside ? acc : 0.0f;
        }
    }
}

// The same two contractions as slots_to_x / x_to_slots_tangent with the grid part read from the Jacobian kept by
// grid_slots(jstore): g[d] += sum_c dl[c] J[d][c] / (2 df)   and   tin[c] = sum_d n[d] J[d][c] / (2 df).
template <int L, int C>
__device__ __forceinline__ void slots_to_x_jac(float divide_factor, const float* jstore, int h, const float (&in)[SDF_IN_STEPS],
                                               const float (&dl)[3 * 16], float (&g)[3]) {
    g[0] = h ? 0.0f : dl[0];
    g[1] = h ? 0.0f : dl[1];
    g[2] = h ? dl[0] : 0.0f;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int g0 = 2 * j, g1 = 2 * j + 1;
        const float sc = h ? (float)(1 << (g1 / 3)) : (float)(1 << (g0 / 3));
        const float t = sc * (in[3 + 2 * j] * dl[2 + 2 * j] - in[2 + 2 * j] * dl[3 + 2 * j]);
        g[g0 % 3] += h ? 0.0f : t;
        g[g1 % 3] += h ? t : 0.0f;
    }
    const float chain = 1.0f / (2.0f * divide_factor);
#pragma unroll
    for (int jl = 0; jl < L / 2; ++jl)
#pragma unroll
        for (int gd = 0; gd < 3; ++gd) {
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) acc = fmaf(dl[20 + jl * C + c], jstore[((jl * 3 + gd) * C + c) * 64], acc);
            g[gd] += acc * chain;
        }
}

template <int L, int C>
__device__ __forceinline__ void tangent_from_jac(float divide_factor, const float* jstore, int h, const float (&in)[SDF_IN_STEPS],
                                                 const float (&n)[3], const float (&dl)[3 * 16], float (&tin)[SDF_IN_STEPS],
                                                 float (&xbar)[3]) {
    tin[0] = h ? n[2] : n[0];
    tin[1] = h ? 0.0f : n[1];
    xbar[0] = xbar[1] = xbar[2] = 0.0f;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int g0 = 2 * j, g1 = 2 * j + 1;
        const float sc = h ? (float)(1 << (g1 / 3)) : (float)(1 << (g0 / 3));
        const float nd = h ? n[g1 % 3] : n[g0 % 3];
        const float s = in[2 + 2 * j], c = in[3 + 2 * j];
        tin[2 + 2 * j] = sc * c * nd;
        tin[3 + 2 * j] = -sc * s * nd;
        const float t = -sc * sc * (s * dl[2 + 2 * j] + c * dl[3 + 2 * j]) * nd;
        xbar[g0 % 3] += h ? 0.0f : t;
        xbar[g1 % 3] += h ? t : 0.0f;
    }
    const float chain = 1.0f / (2.0f * divide_factor);
#pragma unroll
    for (int jl = 0; jl < L / 2; ++jl)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float acc = 0.0f;
#pragma unroll
            for (int gd = 0; gd < 3; ++gd) acc = fmaf(n[gd] * chain, jstore[((jl * 3 + gd) * C + c) * 64], acc);
            tin[20 + jl * C + c] = acc;
        }
}

// Table gradient of one SDF grid for THIS lane's levels (mapping): per corner row,
//   gT[row, c] += w_corner * hb[c]  +  k_corner * dl[c]
// first term = value path (hb = cotangent of the grid features, kernel_grid_backward hashencoder.cu:286-373), second =
// the table's share of the double backward through grad sdf (k_corner = sum_d n_d d w_corner/d x_d, dl = reverse-pass
// vector; kernel_grad2_embeddings :461-625).  Consecutive samples of a ray share cells on the coarse levels, so the
// atomics are run-merged across the half-wave (grid_common.hpp::scatter_runs; keys are table-global rows, which keeps
// the two halves -- different levels -- apart).
template <int L, int C>
__device__ __forceinline__ void table_grad_scatter(const float (&x)[3], float divide_factor, const GridGeom16& geom, int h,
                                                   int lane, bool live, const float (&hb)[3 * 16], const float (&dl)[3 * 16],
                                                   const float (&n)[3], float* __restrict__ g_table, float* lds_tile) {
    float u[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) u[d] = to_unit(x[d], divide_factor);
    const float chain = 1.0f / (2.0f * divide_factor);
#pragma unroll
    for (int jl = 0; jl < L / 2; ++jl) {
        const LevelGeom lg = geom.lv[2 * jl + h];
        uint32_t cell[3];
        float w[3], dw[3];
        const bool active = locate<3>(u, lg.scale, cell, w, dw) && live;
        float k[8];
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) k[corner] = 0.0f;
#pragma unroll
        for (int gd = 0; gd < 3; ++gd) {
#pragma unroll
            for (int face = 0; face < 4; ++face) {
                float wt = lg.scale;
                int lo = 0;
#pragma unroll
                for (int nd = 0; nd < 2; ++nd) {
                    const int d = (nd >= gd) ? nd + 1 : nd;
                    if ((face >> nd) & 1) { wt *= w[d]; lo |= 1 << d; }
                    else                  { wt *= 1.0f - w[d]; }
                }
                const float t = wt * dw[gd] * n[gd] * chain;
                k[lo | (1 << gd)] += t;
                k[lo] -= t;
            }
        }
        // both levels of this iteration dense (uniform across the wave): x-neighbour corners go out as row pairs
        uint32_t row[8];
        float wt8[8];
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            float wt = 1.0f;
            uint32_t q[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int bit = (corner >> d) & 1;
                wt *= bit ? w[d] : 1.0f - w[d];
                q[d] = cell[d] + bit;
            }
            row[corner] = lg.row0 + level_row<3>(lg, q);
            wt8[corner] = wt;
        }
#pragma unroll
        for (int yz = 0; yz < 4; ++yz) {
            float v0[C], v1[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                v0[c] = fmaf(wt8[2 * yz], hb[20 + jl * C + c], k[2 * yz] * dl[20 + jl * C + c]);
                v1[c] = fmaf(wt8[2 * yz + 1], hb[20 + jl * C + c], k[2 * yz + 1] * dl[20 + jl * C + c]);
            }
            scatter_x_pair<C>(g_table, row[2 * yz], row[2 * yz + 1], active, v0, v1, lane, lds_tile);
        }
    }
}

// Where a point comes from: sample `pid % S` of ray `pid / S` (x = o + z d), or an explicit point list.
struct PointSrc {
    const float* rays_o;   // [R,3]
    const float* rays_d;   // [R,3]
    const float* z_vals;   // [R,S]
    const float* points;   // [P,3] or nullptr
    uint32_t P, S;
    const uint32_t* order; // optional launch order: work item i handles point order[i] (spatially sorted, see
                           // nsa_morton_keys); tile-indexed buffers (HL, save, emission) follow the work items
};

__device__ __forceinline__ uint32_t point_of(const PointSrc& ps, uint32_t work_item) {
    return ps.order ? ps.order[work_item] : work_item;
}

__device__ __forceinline__ void load_point(const PointSrc& ps, uint32_t pid, float (&x)[3], uint32_t& ray, float& z) {
    if (ps.points) {
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = ps.points[(size_t)pid * 3 + k];
        ray = 0;
        z = 0.0f;
        return;
    }
    ray = pid / ps.S;
    z = ps.z_vals[pid];
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = ps.rays_o[ray * 3 + k] + mul_rn(z, ps.rays_d[ray * 3 + k]);
}

}  // namespace nsa
