// sdf_net4.hpp -- the SDF networks (coarse 71->64->65, fine 71->64->64->64->65; Softplus(100)) in the quad tiling of mlp16.hpp:
// a wave = 16 points, four lanes per point, fused with their grid encoders and the positional encoding.
// Reference: ImplicitNetworkGrid.forward / get_outputs (code/model/base_networks.py:155-221), HashEncoder.forward
// (code/hashencoder/hashgrid.py:199-215), Embedder (code/model/embedder.py:5-37).
//
// First-layer slots of quarter-lane q (24 per lane = 3 k-groups; reference feature index in brackets; k = frequency 2^k, d =
// coordinate).  The map keeps every coordinate index a compile-time constant or a two-way choice (a lane-varying three-way
// choice compiles to divergent branches):
//   slots 2n, 2n+1, n < 3  : sin, cos of 2^q x_n                                  [3+6q+n], [6+6q+n]
//   slots 6, 7             : sin, cos of 2^(4+(q&1)) x_d, d = (q & 2) ? 2 : 0    [3+6k+d], [6+6k+d]
//   slots 8, 9             : q < 2: sin, cos of 2^(4+q) x_1   [27+6q+1], [30+6q+1] ;  q == 3: x_0, x_1   [0], [1] ;  q == 2: zero
//   slot 10                : q == 3: x_2   [2] ;  else zero
//   slots 11..15           : zero pad
//   slots 16 + jl*C + c    : channel c of grid level q + 4 jl   [39 + (q + 4 jl) C + c]        (jl < 8/C: coarse 1, fine 2)
// (nicer_slam_amd/fused/pack.py::sdf_in_feature4 builds the packed weights from exactly this table; with level = q + 4 jl
//  the four quarters of a wave work on levels 0..3 together, then 4..7 -- the dense levels first.)
#pragma once
#include "mlp16.hpp"
#include "sdf_net.hpp"      // GridGeom16, PointSrc, load_point, scatter primitives (grid_common.hpp)

namespace nsa {

// Offsets (in floats) inside one SDF net's packed parameter block, quad layout; NH = number of hidden layers.
template <int NH>
struct SdfPack4 {
    static constexpr int kHH = a16_floats(4, 2);                      // one 64x64 hidden block: A[4 tiles][2 groups]
    static constexpr int kW0 = 0;                                     // A[4][3]
    static constexpr int kB0 = kW0 + a16_floats(4, QIN_G);
    static constexpr int kWH = kB0 + 64;                              // (NH-1) x { A[4][2], bias[64] }
    static constexpr int kWSDF = kWH + (NH - 1) * (kHH + 64);         // last-layer row 0 in activation layout
    static constexpr int kBSDF = kWSDF + 64;                          // [0] = bias of the sdf output
    static constexpr int kWFEAT = kBSDF + 64;                         // last-layer rows 1..64: A[4][2]
    static constexpr int kBFEAT = kWFEAT + kHH;
    static constexpr int kWHT = kBFEAT + 64;                          // transposed hidden layers, order k = NH-1 .. 1
    static constexpr int kW0T = kWHT + (NH - 1) * kHH;                // A[6 tiles][2 groups]: rows = first-layer slots
    static constexpr int kWFEATT = kW0T + a16_floats(6, 2);           // transposed feature rows
    static constexpr int kTotal = kWFEATT + kHH;
    __host__ __device__ static constexpr int wh(int k) { return kWH + (k - 1) * (kHH + 64); }       // k = 1..NH-1
    __host__ __device__ static constexpr int bh(int k) { return wh(k) + kHH; }
    __host__ __device__ static constexpr int wht(int k) { return kWHT + (NH - 1 - k) * kHH; }       // k = 1..NH-1
};

// level geometry in LDS: a quarter-lane picks its level with a per-lane index (the kernel-argument copy would need a
// 4-way select per field)
__device__ __forceinline__ void geom_to_lds(const GridGeom16& geom, LevelGeom* s_geom) {
    if (threadIdx.x < 16) s_geom[threadIdx.x] = geom.lv[threadIdx.x];
}

// a quarter-lane's level record out of LDS.  NSA_X_GEOM_VOLATILE (SLP-hazard bisect, experiment builds only): field by field through
// volatile reads, so that no two fields can be merged into one ds_read2_b32.
__device__ __forceinline__ LevelGeom load_geom(const LevelGeom* p) {
#ifdef NSA_X_GEOM_VOLATILE
    const volatile uint32_t* w = reinterpret_cast<const volatile uint32_t*>(p);
    uint32_t r[sizeof(LevelGeom) / 4];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(LevelGeom) / 4); ++i) r[i] = w[i];
    LevelGeom g;
    __builtin_memcpy(&g, r, sizeof(LevelGeom));
    return g;
#else
    return *p;
#endif
}

__device__ __forceinline__ float pow2f(int k) { return __uint_as_float((uint32_t)(127 + k) << 23); }

// coordinate and frequency of positional-encoding pair n (slots 2n, 2n+1) of quarter q; sc == 0: the pair does not exist
struct PairSel {
    float xa;     // the coordinate (or, for the tangent, the cotangent component) it reads
    float sc;     // 2^k
};
__device__ __forceinline__ PairSel pe_pair(int n, int q, const float (&x)[3]) {
    PairSel r;
    if (n < 3) { r.xa = x[n]; r.sc = pow2f(q); }
    else if (n == 3) { r.xa = (q & 2) ? x[2] : x[0]; r.sc = pow2f(4 + (q & 1)); }
    else { r.xa = x[1]; r.sc = q < 2 ? pow2f(4 + q) : 0.0f; }
    return r;
}
// g[d(n, q)] += t
__device__ __forceinline__ void pe_add(int n, int q, float (&g)[3], float t) {
    if (n < 3) g[n] += t;
    else if (n == 3) { g[0] += (q & 2) ? 0.0f : t; g[2] += (q & 2) ? t : 0.0f; }
    else g[1] += t;                     // (t carries sc == 0 for the quarters without this pair)
}

// Position + positional-encoding slots 0..15 (identical for the coarse and the fine network).
__device__ __forceinline__ void pe_slots4(const float (&x)[3], int q, float (&in)[QIN]) {
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        const PairSel ps = pe_pair(n, q, x);
        float s, c;
        sincos_f(ps.xa * ps.sc, s, c);
        in[2 * n] = s;
        in[2 * n + 1] = c;
    }
    // slots 8, 9: the pair exists for q < 2 only; quarter 3 keeps x_0, x_1 there and x_2 in slot 10
    in[8] = q < 2 ? in[8] : (q == 3 ? x[0] : 0.0f);
    in[9] = q < 2 ? in[9] : (q == 3 ? x[1] : 0.0f);
    in[10] = q == 3 ? x[2] : 0.0f;
#pragma unroll
    for (int s = 11; s < 16; ++s) in[s] = 0.0f;
}

// Grid-feature slots 16..23: the 8/C levels (q + 4 jl) of this lane, C channels each.  jstore != nullptr: also keeps
// d feature / d u (3 x C Jacobian rows per level, zero outside the grid) in a lane-private LDS column,
// jstore[((jl*3 + d)*C + c) * 64].
template <int L, int C>
__device__ __forceinline__ void grid_slots4(const float (&x)[3], float divide_factor, const float* __restrict__ table,
                                            const LevelGeom* s_geom, int q, float (&in)[QIN], float* jstore = nullptr) {
    static_assert(L * C == 32 && (C == 4 || C == 8), "quad layout: 32 grid features, 8 per quarter-lane");
    float u[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) u[d] = to_unit(x[d], divide_factor);   // hashgrid.py:203 (size = 1)
#pragma unroll
    for (int jl = 0; jl < 8 / C; ++jl) {
        const LevelGeom g = load_geom(&s_geom[q + 4 * jl]);
        uint32_t cell[3];
        float w[3], dw[3];
        const bool inside = locate<3>(u, g.scale, cell, w, dw);
        float v[8][C];
        gather_corners<3, C>(table, g, cell, v);
        float f[C];
        blend<3, C>(v, w, f);
#pragma unroll
        for (int c = 0; c < C; ++c) in[16 + jl * C + c] = inside ? f[c] : 0.0f;
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

// SDF value only (sampler pass): acc chain through the hidden layers, then the sdf row as a VALU dot.
// GEMM = functor (wp_offset, KG/MT as template args) supplied by the kernel: staged-from-LDS or streamed.
template <int NH, class Gemm>
__device__ __forceinline__ float sdf_only4(const float* __restrict__ wp, int q, const float (&in)[QIN], Gemm& gemm) {
    using P = SdfPack4<NH>;
    f32x4v acc[4];
    load_vec16(wp + P::kB0, q, acc);
    gemm.template run<QIN_G, 4>(P::kW0, in, acc);
    float act[QHS];
#pragma unroll
    for (int k = 1; k < NH; ++k) {
        const float bound = acc_abs_max16<4>(acc) + kSoftplusSlack;      // (form 2: the next GEMM's scale, known before its operands)
#pragma unroll
        for (int s = 0; s < QHS; ++s) act[s] = softplus100(acc[s >> 2][s & 3]);
        load_vec16(wp + P::bh(k), q, acc);
        gemm.template run<2, 4>(P::wh(k), act, acc, &bound);
    }
    f32x4v ws[4];
    load_vec16(wp + P::kWSDF, q, ws);
    float part = 0.0f;
#pragma unroll
    for (int s = 0; s < QHS; ++s) part = fmaf(softplus100(acc[s >> 2][s & 3]), ws[s >> 2][s & 3], part);
    return quad_sum(part) + wp[P::kBSDF];
}

// ------------------------------------------------------------------------------------------------------------------
// Contraction of a per-slot cotangent with d(first-layer input)/dx for THIS lane's share of the slots:
//   g[d] = sum_slots dl[slot] * d in[slot] / d x_d.   Caller adds the four quarter-lanes (quad_sum).
__device__ __forceinline__ void pe_to_x4(int q, const float (&in)[QIN], const float (&dl)[QIN], float (&g)[3]) {
    g[0] = q == 3 ? dl[8] : 0.0f;
    g[1] = q == 3 ? dl[9] : 0.0f;
    g[2] = q == 3 ? dl[10] : 0.0f;
    const float zero3[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        const float sc = pe_pair(n, q, zero3).sc;
        pe_add(n, q, g, sc * (in[2 * n + 1] * dl[2 * n] - in[2 * n] * dl[2 * n + 1]));   // 2^k (cos d_sin - sin d_cos)
    }
}

// grid part by re-gathering the corner rows (L2-resident SDF tables): sum_faces w_face * (p_hi - p_lo), p_corner = <dl_level, row>
template <int L, int C>
__device__ __forceinline__ void slots_to_x4(const float (&x)[3], float divide_factor, const float* __restrict__ table,
                                            const LevelGeom* s_geom, int q, const float (&in)[QIN], const float (&dl)[QIN],
                                            float (&g)[3]) {
    pe_to_x4(q, in, dl, g);
    float u[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) u[d] = to_unit(x[d], divide_factor);
    const float chain = 1.0f / (2.0f * divide_factor);
#pragma unroll
    for (int jl = 0; jl < 8 / C; ++jl) {
        const LevelGeom lg = load_geom(&s_geom[q + 4 * jl]);
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
            for (int c = 0; c < C; ++c) p[corner] = fmaf(dl[16 + jl * C + c], v[corner][c], p[corner]);
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

// Tangent of the first-layer input along n (a cotangent of grad sdf): tin[slot] = sum_d (d in[slot]/d x_d) n_d -- the grid
// Hessian term is NOT part of it (hashgrid.py:134) -- and the positional-encoding second-derivative term
//   xbar_d += n_d * sum_k -(4^k) (sin * dl_sin + cos * dl_cos).
__device__ __forceinline__ void pe_tangent4(int q, const float (&in)[QIN], const float (&n)[3], const float (&dl)[QIN],
                                            float (&tin)[QIN], float (&xbar)[3]) {
    xbar[0] = xbar[1] = xbar[2] = 0.0f;
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const PairSel ps = pe_pair(m, q, n);                 // xa = the cotangent component n_d of this pair's coordinate
        const float s = m == 4 && q >= 2 ? 0.0f : in[2 * m], c = m == 4 && q >= 2 ? 0.0f : in[2 * m + 1];   // (q == 3 keeps x there)
        tin[2 * m] = ps.sc * c * ps.xa;
        tin[2 * m + 1] = -ps.sc * s * ps.xa;
        pe_add(m, q, xbar, -ps.sc * ps.sc * (s * dl[2 * m] + c * dl[2 * m + 1]) * ps.xa);
    }
    tin[8] = q == 3 ? n[0] : tin[8];
    tin[9] = q == 3 ? n[1] : tin[9];
    tin[10] = q == 3 ? n[2] : 0.0f;
#pragma unroll
    for (int s = 11; s < 16; ++s) tin[s] = 0.0f;
}

template <int L, int C>
__device__ __forceinline__ void x_to_slots_tangent4(const float (&x)[3], float divide_factor, const float* __restrict__ table,
                                                    const LevelGeom* s_geom, int q, const float (&in)[QIN], const float (&n)[3],
                                                    const float (&dl)[QIN], float (&tin)[QIN], float (&xbar)[3]) {
    pe_tangent4(q, in, n, dl, tin, xbar);
    float u[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) u[d] = to_unit(x[d], divide_factor);
    const float chain = 1.0f / (2.0f * divide_factor);
#pragma unroll
    for (int jl = 0; jl < 8 / C; ++jl) {
        const LevelGeom lg = load_geom(&s_geom[q + 4 * jl]);
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
            tin[16 + jl * C + c] = inside ? acc : 0.0f;
        }
    }
}

// The same two contractions with the grid part read from the Jacobian kept by grid_slots4(jstore).
template <int L, int C>
__device__ __forceinline__ void slots_to_x_jac4(float divide_factor, const float* jstore, int q, const float (&in)[QIN],
                                                const float (&dl)[QIN], float (&g)[3]) {
    pe_to_x4(q, in, dl, g);
    const float chain = 1.0f / (2.0f * divide_factor);
#pragma unroll
    for (int jl = 0; jl < 8 / C; ++jl)
#pragma unroll
        for (int gd = 0; gd < 3; ++gd) {
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) acc = fmaf(dl[16 + jl * C + c], jstore[((jl * 3 + gd) * C + c) * 64], acc);
            g[gd] += acc * chain;
        }
}

template <int L, int C>
__device__ __forceinline__ void tangent_from_jac4(float divide_factor, const float* jstore, int q, const float (&in)[QIN],
                                                  const float (&n)[3], const float (&dl)[QIN], float (&tin)[QIN],
                                                  float (&xbar)[3]) {
    pe_tangent4(q, in, n, dl, tin, xbar);
    const float chain = 1.0f / (2.0f * divide_factor);
#pragma unroll
    for (int jl = 0; jl < 8 / C; ++jl)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float acc = 0.0f;
#pragma unroll
            for (int gd = 0; gd < 3; ++gd) acc = fmaf(n[gd] * chain, jstore[((jl * 3 + gd) * C + c) * 64], acc);
            tin[16 + jl * C + c] = acc;
        }
}

// Table gradient of one SDF grid for THIS lane's levels (mapping): per corner row,
//   gT[row, c] += w_corner * hb[c]  +  k_corner * dl[c]
// (value path, kernel_grid_backward hashencoder.cu:286-373, + the table's share of the double backward through grad sdf,
// kernel_grad2_embeddings :461-625).  The 16 lanes of a quarter are consecutive samples of a ray on ONE level: the run merge
// of scatter_span combines equal rows across them; quarters never merge (keys are table-global rows of different levels).
template <int L, int C>
__device__ __forceinline__ void table_grad_scatter4(const float (&x)[3], float divide_factor, const LevelGeom* s_geom, int q,
                                                    int lane, bool live, const float (&hb)[QIN], const float (&dl)[QIN],
                                                    const float (&n)[3], float* __restrict__ g_table, float* lds_tile) {
    float u[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) u[d] = to_unit(x[d], divide_factor);
    const float chain = 1.0f / (2.0f * divide_factor);
#pragma unroll
    for (int jl = 0; jl < 8 / C; ++jl) {
        const LevelGeom lg = load_geom(&s_geom[q + 4 * jl]);
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
        // x-neighbour corners go out as row pairs wherever they are adjacent rows (dense levels; even cells of hashed ones)
        uint32_t row[8];
        float wt8[8];
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            float wt = 1.0f;
            uint32_t qq[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int bit = (corner >> d) & 1;
                wt *= bit ? w[d] : 1.0f - w[d];
                qq[d] = cell[d] + bit;
            }
            row[corner] = lg.row0 + level_row<3>(lg, qq);
            wt8[corner] = wt;
        }
#pragma unroll
        for (int yz = 0; yz < 4; ++yz) {
            float v0[C], v1[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                v0[c] = fmaf(wt8[2 * yz], hb[16 + jl * C + c], k[2 * yz] * dl[16 + jl * C + c]);
                v1[c] = fmaf(wt8[2 * yz + 1], hb[16 + jl * C + c], k[2 * yz + 1] * dl[16 + jl * C + c]);
            }
            scatter_x_pair<C>(g_table, row[2 * yz], row[2 * yz + 1], active, v0, v1, lane, lds_tile);
        }
    }
}

// Per-point feature vectors travel between kernels in the "HL" layout of the 32-point tiling (render_sdfnet.hip, the colour
// kernels): float index ((tile32*32 + q32)*64 + lane32).  For the quad layout's lane (j, q) of 16-point tile `tile16` and
// activation index s this is   hl_base(...) + hl_step(s):  each register store covers four 64-byte segments.
__device__ __forceinline__ size_t hl_base4(uint32_t tile16, int j, int q) {
    const uint32_t tile32 = tile16 >> 1, p32 = (tile16 & 1) * 16 + j;
    return ((size_t)tile32 * 32 + 4 * (q >> 1)) * 64 + p32 + 32 * (q & 1);
}
__host__ __device__ constexpr int hl_step4(int s) { return (16 * (s >> 3) + (s & 3) + 8 * ((s >> 2) & 1)) * 64; }

}  // namespace nsa
