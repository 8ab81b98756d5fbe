// hash_encode.hip -- gfx950 kernels behind Section 1 of include/nicer_slam_amd.h: the stand-alone
// multi-resolution grid encoder with the reference's operator interface
// (reference: code/hashencoder/src/hashencoder.cu; entry points :758-854).
//
// Mapping (all kernels): 1-D grid, 256-thread blocks (4 waves); a block owns a tile of 64 POINTS and all their levels: lane =
// point, wave w works on levels w, w+4, ...  One lane owns one (point, level) at a time: it gathers the 2^D corner rows with
// 8/16/32-byte vector loads (C = 2/4/8), keeps them in registers and derives both the blended feature and the D Jacobian rows
// from the same values.  Everything that does not depend on the point (scale, strides, hashed/dense, modulo form) arrives as
// kernel arguments and is wave-uniform.
//
// Why a point tile and not (level = block % L), which pins each level's table to the L2s of XCDs {level % 8}: measured on
// MI355X with 802 816 ray-ordered points (tools/bench_hashenc.py, PMC in profiles/r02_hashenc_*): dy_dx is [B][L][D][C], so
// the D*C floats of one (point, level) are a 24..96-byte piece of the point's L*D*C*4-byte row; from a level-pinned mapping those
// pieces arrive from different XCDs (different L2s) and leave as partial 64-byte writes -- 720 MB written for 411 MB of results
// on the colour grid (1 GiB, 16 levels), +416 us over the Jacobian-free forward -- and the seven hashed 128-MiB levels keep
// seven of the eight XCDs waiting on DRAM while the eighth idles.  With point tiles the Jacobian rows are collected in LDS
// ([point][L*D*C], pitch +1) and leave (resp. arrive, second backward) as the points' complete contiguous rows, the point is
// loaded once instead of L times, and every XCD sees the same mix of levels: forward + Jacobian 1283 -> 613 us (colour),
// 324 -> 156 us (fine SDF grid), 161 -> 143 us (coarse).
#include "grid_common.hpp"

namespace nsa {

constexpr int TPB = 256;
constexpr int NWAVE = TPB / 64;
constexpr uint32_t TILE = 64;          // points per block

// coalesced copy of the tile's complete dy_dx rows between global memory and the LDS tile [TILE][row + 1]
template <bool TO_LDS>
__device__ __forceinline__ void jac_rows_copy(float* __restrict__ global_rows, float* tile, uint32_t npts, uint32_t row,
                                              int lane, int wave) {
    for (uint32_t pt = wave; pt < npts; pt += NWAVE) {
        float* g = global_rows + (size_t)pt * row;
        float* t = tile + pt * (row + 1);
        for (uint32_t k = lane; k < row; k += 64) {
            if (TO_LDS) t[k] = g[k];
            else        g[k] = t[k];
        }
    }
}

// ------------------------------------------------------------------------------------------ forward
// `stage`: the dy_dx tile fits the LDS allocation of this launch (always, unless L*D*C is unusually large)
template <int D, int C, bool JAC>
__global__ __launch_bounds__(TPB) void k_grid_forward(const float* __restrict__ inputs, const float* __restrict__ emb,
                                                      float* __restrict__ outputs, float* __restrict__ dy_dx,
                                                      uint32_t B, uint32_t L, GridGeom geom, int stage) {
    extern __shared__ float jac_tile[];                       // [TILE][L*D*C + 1] when JAC && stage
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t p0 = blockIdx.x * TILE;
    const uint32_t b = p0 + lane;
    const bool have = b < B;
    const uint32_t row = L * D * C;
    float x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = have ? inputs[(size_t)b * D + d] : -1.0f;
    for (uint32_t level = wave; level < L; level += NWAVE) {
        const LevelGeom g = geom.lv[level];
        uint32_t cell[D];
        float w[D], dw[D];
        const bool inside = locate<D>(x, g.scale, cell, w, dw) && have;
        float out[C], j[D][C];
#pragma unroll
        for (int c = 0; c < C; ++c) out[c] = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int c = 0; c < C; ++c) j[d][c] = 0.0f;
        if (inside) {            // zeros outside the unit cube: hashencoder.cu:161-177
            float v[1 << D][C];
            gather_corners<D, C, true>(emb, g, cell, v);
            blend<D, C>(v, w, out);
            if (JAC) {
#pragma unroll
                for (int gd = 0; gd < D; ++gd) jacobian_row<D, C>(v, w, dw, g.scale, gd, j[gd]);
            }
        }
        if (have) store_row<C>(outputs + ((size_t)level * B + b) * C, out);
        if (JAC) {
            if (stage) {
                float* t = jac_tile + lane * (row + 1) + level * D * C;
#pragma unroll
                for (int d = 0; d < D; ++d)
#pragma unroll
                    for (int c = 0; c < C; ++c) t[d * C + c] = j[d][c];
            } else if (have) {
#pragma unroll
                for (int d = 0; d < D; ++d) store_row<C>(dy_dx + ((size_t)b * L + level) * D * C + d * C, j[d]);
            }
        }
    }
    if (JAC && stage) {
        __syncthreads();
        jac_rows_copy<false>(dy_dx + (size_t)p0 * row, jac_tile, B - p0 < TILE ? B - p0 : TILE, row, lane, wave);
    }
}

// -------------------------------------------------------------------- first backward: table scatter
// grad_emb[row(corner), c] += w(corner) * grad[l,b,c]   (kernel_grid_backward :286-373)
template <int D, int C>
__global__ __launch_bounds__(TPB) void k_grid_scatter(const float* __restrict__ grad, const float* __restrict__ inputs,
                                                      float* __restrict__ grad_emb, uint32_t B, uint32_t L, GridGeom geom) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * TILE + lane;
    const bool have = b < B;                       // no early exit: every lane takes part in the run merge
    float x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = have ? inputs[(size_t)b * D + d] : -1.0f;
    for (uint32_t level = wave; level < L; level += NWAVE) {
        const LevelGeom g = geom.lv[level];
        uint32_t cell[D] = {};               // (lanes without a point skip locate(): keep the row arithmetic below defined)
        float w[D], dw[D];
        const bool active = have && locate<D>(x, g.scale, cell, w, dw);     // out-of-range points add nothing (:313-317)
        float gy[C];
#pragma unroll
        for (int c = 0; c < C; ++c) gy[c] = 0.0f;
        if (active) load_row<C>(grad + ((size_t)level * B + b) * C, gy);
        float* tl = grad_emb + (size_t)g.row0 * C;
        // corners in x-neighbour pairs (2 yz, 2 yz + 1): one span per pair where the two rows are adjacent (grid_common.hpp)
#pragma unroll
        for (int yz = 0; yz < (1 << (D - 1)); ++yz) {
            uint32_t r[2];
            float v[2][C];
#pragma unroll
            for (int xb = 0; xb < 2; ++xb) {
                const int corner = 2 * yz + xb;
                float wt = 1.0f;
                uint32_t q[D];
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int bit = (corner >> d) & 1;
                    wt *= bit ? w[d] : 1.0f - w[d];
                    q[d] = cell[d] + bit;
                }
                r[xb] = level_row<D>(g, q);
#pragma unroll
                for (int c = 0; c < C; ++c) v[xb][c] = wt * gy[c];
            }
            scatter_x_pair<C>(tl, r[0], r[1], active, v[0], v[1], lane);
        }
    }
}

// -------------------------------------------------------------- first backward: input gradient J^T g
// grad_inputs[b,d] = sum_l sum_c grad[l,b,c] * dy_dx[b,l,d,c]   (kernel_input_backward :376-402)
// One thread per point: its dy_dx row (L*D*C floats) is contiguous and read with 16-byte loads.
template <int D, int C>
__global__ __launch_bounds__(TPB) void k_input_backward(const float* __restrict__ grad, const float* __restrict__ dy_dx,
                                                        float* __restrict__ grad_inputs, uint32_t B, uint32_t L) {
    const uint32_t b = blockIdx.x * TPB + threadIdx.x;
    if (b >= B) return;
    float acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.0f;
    const float* jrow = dy_dx + (size_t)b * L * D * C;
    for (uint32_t l = 0; l < L; ++l) {
        float gy[C];
        load_row<C>(grad + ((size_t)l * B + b) * C, gy);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float j[C];
            load_row<C>(jrow + ((size_t)l * D + d) * C, j);
#pragma unroll
            for (int c = 0; c < C; ++c) acc[d] += gy[c] * j[c];
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) grad_inputs[(size_t)b * D + d] = acc[d];
}

// ------------------------------------------------------------------------------- second backward
// grad_grad[l,b,c] = sum_d ggi[b,d]*dy_dx[b,l,d,c]                 (:405-458)
// grad2_emb[row(corner),c] += sum_gd +-scale*w_{-gd}*grad[l,b,c]*ggi[b,gd]*smoothstep'(t_gd)   (:461-625)
// Fused into one pass over (point, level); the d/dx of J^T g is NOT produced (hashgrid.py:134).
template <int D, int C, bool SCATTER>
__global__ __launch_bounds__(TPB) void k_grid_second_backward(const float* __restrict__ grad, const float* __restrict__ inputs,
                                                              const float* __restrict__ dy_dx, const float* __restrict__ ggi_,
                                                              float* __restrict__ grad_grad, float* __restrict__ grad2_emb,
                                                              uint32_t B, uint32_t L, GridGeom geom, int stage) {
    extern __shared__ float jac_tile[];                       // [TILE][L*D*C + 1] when stage
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t p0 = blockIdx.x * TILE;
    const uint32_t b = p0 + lane;
    const bool have = b < B;
    const uint32_t row = L * D * C;
    if (stage) {
        jac_rows_copy<true>(const_cast<float*>(dy_dx) + (size_t)p0 * row, jac_tile, B - p0 < TILE ? B - p0 : TILE, row, lane, wave);
        __syncthreads();
    }
    float ggi[D], x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ggi[d] = have ? ggi_[(size_t)b * D + d] : 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = have && SCATTER ? inputs[(size_t)b * D + d] : -1.0f;
    for (uint32_t level = wave; level < L; level += NWAVE) {
        const LevelGeom g = geom.lv[level];
        if (have) {
            float r[C];
#pragma unroll
            for (int c = 0; c < C; ++c) r[c] = 0.0f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                float j[C];
                if (stage) {
                    const float* t = jac_tile + lane * (row + 1) + (level * D + d) * C;
#pragma unroll
                    for (int c = 0; c < C; ++c) j[c] = t[c];
                } else {
                    load_row<C>(dy_dx + (((size_t)b * L + level) * D + d) * C, j);
                }
#pragma unroll
                for (int c = 0; c < C; ++c) r[c] += ggi[d] * j[c];
            }
            store_row<C>(grad_grad + ((size_t)level * B + b) * C, r);
        }
        if (!SCATTER) continue;
        uint32_t cell[D] = {};
        float w[D], dw[D];
        const bool active = have && locate<D>(x, g.scale, cell, w, dw);
        float gy[C];
#pragma unroll
        for (int c = 0; c < C; ++c) gy[c] = 0.0f;
        if (active) load_row<C>(grad + ((size_t)level * B + b) * C, gy);
        // per-corner scalar coefficient k[corner] = sum_gd sign * scale * prod_{d != gd} w_d * ggi[gd] * dw[gd]
        float k[1 << D];
#pragma unroll
        for (int corner = 0; corner < (1 << D); ++corner) k[corner] = 0.0f;
#pragma unroll
        for (int gd = 0; gd < D; ++gd) {
#pragma unroll
            for (int face = 0; face < (1 << (D - 1)); ++face) {
                float wt = g.scale;
                int lo = 0;
#pragma unroll
                for (int nd = 0; nd < D - 1; ++nd) {
                    const int d = (nd >= gd) ? nd + 1 : nd;
                    if ((face >> nd) & 1) { wt *= w[d]; lo |= 1 << d; }
                    else                  { wt *= 1.0f - w[d]; }
                }
                const float t = wt * ggi[gd] * dw[gd];
                k[lo | (1 << gd)] += t;
                k[lo] -= t;
            }
        }
        float* tl = grad2_emb + (size_t)g.row0 * C;
#pragma unroll
        for (int yz = 0; yz < (1 << (D - 1)); ++yz) {
            uint32_t r[2];
            float v[2][C];
#pragma unroll
            for (int xb = 0; xb < 2; ++xb) {
                const int corner = 2 * yz + xb;
                uint32_t q[D];
#pragma unroll
                for (int d = 0; d < D; ++d) q[d] = cell[d] + ((corner >> d) & 1);
                r[xb] = level_row<D>(g, q);
#pragma unroll
                for (int c = 0; c < C; ++c) v[xb][c] = k[corner] * gy[c];
            }
            scatter_x_pair<C>(tl, r[0], r[1], active, v[0], v[1], lane);
        }
    }
}

static inline uint32_t tiles(uint32_t B) { return (B + TPB - 1) / TPB; }
static inline uint32_t point_tiles(uint32_t B) { return (B + TILE - 1) / TILE; }
static inline int launch_status() { return launch_end(); }
// LDS bytes of the dy_dx tile, or 0 when it would not fit the default 64 KiB dynamic allocation (the kernels then touch the
// row pieces in global memory directly)
// `static_lds`: what the kernel allocates statically on top (the scatter variant of the second backward keeps scatter_x_pair's
// stage2[4][64 * (2C + 1)] floats): both together must fit the 64 KiB a launch may ask for without opting into more.
static inline size_t jac_tile_bytes(uint32_t L, int D, int C, size_t static_lds = 0) {
    const size_t n = TILE * ((size_t)L * D * C + 1) * sizeof(float);
    return n + static_lds <= 64 * 1024 ? n : 0;
}

template <int D, int C>
static int forward_dc(const float* in, const float* emb, float* out, float* dy_dx, uint32_t B, uint32_t L, bool jac,
                      const GridGeom& geom, hipStream_t st) {
    const dim3 grid(point_tiles(B)), block(TPB);
    const size_t lds = jac ? jac_tile_bytes(L, D, C) : 0;
    launch_begin();
    if (jac) hipLaunchKernelGGL((k_grid_forward<D, C, true>), grid, block, lds, st, in, emb, out, dy_dx, B, L, geom, lds != 0);
    else     hipLaunchKernelGGL((k_grid_forward<D, C, false>), grid, block, 0, st, in, emb, out, dy_dx, B, L, geom, 0);
    return launch_status();
}

template <int D, int C>
static int backward_dc(const float* grad, const float* in, float* gemb, uint32_t B, uint32_t L, bool gi, const float* dy_dx,
                       float* gin, const GridGeom& geom, hipStream_t st) {
    launch_begin();
    if (gemb) hipLaunchKernelGGL((k_grid_scatter<D, C>), dim3(point_tiles(B)), dim3(TPB), 0, st, grad, in, gemb, B, L, geom);
    if (gi) hipLaunchKernelGGL((k_input_backward<D, C>), dim3(tiles(B)), dim3(TPB), 0, st, grad, dy_dx, gin, B, L);
    return launch_status();
}

template <int D, int C>
static int second_dc(const float* grad, const float* in, const float* dy_dx, const float* ggi, float* gg, float* g2emb,
                     uint32_t B, uint32_t L, const GridGeom& geom, hipStream_t st) {
    const dim3 grid(point_tiles(B)), block(TPB);
    // reading: a (point, level) piece of D*C*4 = 96 bytes (C = 8) is fetched whole anyway and the tile only costs occupancy
    // (coarse SDF grid: 327 us direct, 463 us staged); 24- and 48-byte pieces gain from arriving as complete rows
    const size_t lds = D * C * sizeof(float) < 64 ? jac_tile_bytes(L, D, C, g2emb ? 4 * 64 * (2 * C + 1) * sizeof(float) : 0) : 0;
    launch_begin();
    if (g2emb) hipLaunchKernelGGL((k_grid_second_backward<D, C, true>), grid, block, lds, st, grad, in, dy_dx, ggi, gg, g2emb, B, L, geom, lds != 0);
    else       hipLaunchKernelGGL((k_grid_second_backward<D, C, false>), grid, block, lds, st, grad, in, dy_dx, ggi, gg, g2emb, B, L, geom, lds != 0);
    return launch_status();
}

}  // namespace nsa

#define NSA_DISPATCH_DC(D, C, CALL)                                                             \
    do {                                                                                        \
        if ((D) != 2 && (D) != 3) return NSA_EUNSUPPORTED_C; /* same text as the reference */   \
        switch ((C) * 10 + (D)) {                                                               \
            case 12: { constexpr int D_ = 2, C_ = 1; return CALL; }                             \
            case 13: { constexpr int D_ = 3, C_ = 1; return CALL; }                             \
            case 22: { constexpr int D_ = 2, C_ = 2; return CALL; }                             \
            case 23: { constexpr int D_ = 3, C_ = 2; return CALL; }                             \
            case 42: { constexpr int D_ = 2, C_ = 4; return CALL; }                             \
            case 43: { constexpr int D_ = 3, C_ = 4; return CALL; }                             \
            case 82: { constexpr int D_ = 2, C_ = 8; return CALL; }                             \
            case 83: { constexpr int D_ = 3, C_ = 8; return CALL; }                             \
            default: return NSA_EUNSUPPORTED_C;                                                 \
        }                                                                                       \
    } while (0)

extern "C" {

const char* nsa_strerror(int code) {
    switch (code) {
        case NSA_OK: return "ok";
        case NSA_EUNSUPPORTED_C: return "GridEncoding: C must be 1, 2, 4, or 8.";
        case NSA_ETOO_MANY_LEVELS: return "GridEncoding: more than NSA_MAX_LEVELS (32) levels";
        case NSA_ELAUNCH: return "HIP kernel launch failed";
        case NSA_EBADARG: return "bad argument (null pointer or inconsistent sizes)";
        case NSA_EUNSUPPORTED_NET: return "fused render core: network shape outside the compiled set";
        default: return "unknown error";
    }
}

int nsa_version(void) { return 1; }

int nsa_hash_encode_forward(const float* inputs, const float* embeddings, const int32_t* offsets_host, float* outputs,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                            float* dy_dx, nsa_stream_t stream) {
    if (B == 0) return NSA_OK;
    if (!inputs || !embeddings || !offsets_host || !outputs || (calc_grad_inputs && !dy_dx)) return NSA_EBADARG;
    nsa::GridGeom geom;
    if (int rc = nsa::make_grid_geom(offsets_host, L, D, S, H, &geom)) return rc;
    NSA_DISPATCH_DC(D, C, (nsa::forward_dc<D_, C_>(inputs, embeddings, outputs, dy_dx, B, L, calc_grad_inputs != 0, geom,
                                                   (hipStream_t)stream)));
}

int nsa_hash_encode_backward(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets_host,
                             float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             int calc_grad_inputs, const float* dy_dx, float* grad_inputs, nsa_stream_t stream) {
    (void)embeddings;
    if (B == 0) return NSA_OK;
    if (!grad || !inputs || !offsets_host || (calc_grad_inputs && (!dy_dx || !grad_inputs))) return NSA_EBADARG;
    nsa::GridGeom geom;
    if (int rc = nsa::make_grid_geom(offsets_host, L, D, S, H, &geom)) return rc;
    NSA_DISPATCH_DC(D, C, (nsa::backward_dc<D_, C_>(grad, inputs, grad_embeddings, B, L, calc_grad_inputs != 0, dy_dx,
                                                    grad_inputs, geom, (hipStream_t)stream)));
}

int nsa_hash_encode_second_backward(const float* grad, const float* inputs, const float* embeddings,
                                    const int32_t* offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                    uint32_t H, int calc_grad_inputs, const float* dy_dx, const float* grad_grad_inputs,
                                    float* grad_grad, float* grad2_embeddings, nsa_stream_t stream) {
    (void)embeddings; (void)calc_grad_inputs;
    if (C == 1) return NSA_EUNSUPPORTED_C;   // hashencoder.cu:708-714: the C=1 case is commented out
    if (B == 0) return NSA_OK;
    if (!grad || !inputs || !offsets_host || !dy_dx || !grad_grad_inputs || !grad_grad) return NSA_EBADARG;
    nsa::GridGeom geom;
    if (int rc = nsa::make_grid_geom(offsets_host, L, D, S, H, &geom)) return rc;
    NSA_DISPATCH_DC(D, C, (nsa::second_dc<D_, C_>(grad, inputs, dy_dx, grad_grad_inputs, grad_grad, grad2_embeddings, B, L,
                                                  geom, (hipStream_t)stream)));
}

}  // extern "C"
