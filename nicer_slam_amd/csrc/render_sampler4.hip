// render_sampler4.hip -- the SDF-only passes (coarse sampler stage, batch inference) in the QUAD tiling (mlp16.hpp, sdf_net4.hpp):
// a wave = 16 points, four lanes per point.  Same mathematics and buffers as k_sampler_sdf / k_sdf_points of
// render_sampler.hip; selected with nsa_grid_t.tile == 16.
//
// Weight residency: these passes only run the forward chain W0 (coarse) and W0, W1, W2 (fine) -- 120 KiB of packed blocks.
// One persistent workgroup per CU copies them into LDS ONCE (asynchronous global->LDS copies), then its waves loop over
// 16-point tiles with no barrier and no weight traffic at all: the per-wave L2 streaming of the 32-point kernel was 12 % of its
// time (round-1 ablation build), and staging with a barrier per GEMM cost it a wave per SIMD.
// Reference: UniformSampler.get_z_vals (code/model/ray_sampler.py:37-61), ImplicitNetworkGrid_COMBINE.get_sdf_vals
// (code/model/base_networks.py:25-35).
#include "sdf_net4.hpp"

namespace nsa {

#ifndef NSA_NWS4
#define NSA_NWS4 12             // waves per persistent workgroup (3 per SIMD at <= 168 registers)
#endif
constexpr int NWS4 = NSA_NWS4;

constexpr int kLdsCoarseW0 = 0;
constexpr int kLdsFineW0 = a16_floats(4, QIN_G);
constexpr int kLdsFineW1 = kLdsFineW0 + a16_floats(4, QIN_G);
constexpr int kLdsFineW2 = kLdsFineW1 + a16_floats(4, 2);
constexpr int kLdsWeights = kLdsFineW2 + a16_floats(4, 2);          // 30720 floats = 120 KiB

// A blocks come from the resident LDS copy; `base` = LDS offset of the network's first block, the packed-block offsets of the
// hidden layers map onto consecutive LDS blocks
template <int NH>
struct ResidentGemm {
    const float* lds;
    int lane;
    const float* wp;           // the network's packed block in global memory (experiment builds only, see below)
    template <int KG, int MT>
    __device__ __forceinline__ void run(int pack_off, const float (&b)[8 * KG], f32x4v (&acc)[MT], const float* hint = nullptr) {
        using P = SdfPack4<NH>;
#ifdef NSA_X_GLB_WEIGHTS
        // SLP-hazard bisect (tools/slp_bisect.sh, profiles/r05_slp_bisect.txt): the same fragments streamed from global memory --
        // no LDS read sits between the MFMAs.  Experiment builds only (build.py refuses NSA_X_* for the product).
        gemm16_glb<KG, MT>(wp + pack_off, lane, b, acc, hint);
#else
        const int off = pack_off == P::kW0 ? 0 : a16_floats(4, QIN_G) + (pack_off - P::wh(1)) / (P::kHH + 64) * P::kHH;
        gemm16_lds<KG, MT>(lds + off, lane, b, acc, hint);
#endif
    }
};

template <int NHC, int NHF>
__device__ __forceinline__ void load_resident_weights(float* lds_w, const float* __restrict__ wp_c, const float* __restrict__ wp_f) {
    using PC = SdfPack4<NHC>;
    using PF = SdfPack4<NHF>;
    static_assert(NHC == 1 && NHF == 3, "resident layout: coarse W0 | fine W0, W1, W2");
    stage_issue_n<NWS4>(wp_c + PC::kW0, a16_floats(4, QIN_G), lds_w + kLdsCoarseW0);
    if (wp_f) {
        stage_issue_n<NWS4>(wp_f + PF::kW0, a16_floats(4, QIN_G), lds_w + kLdsFineW0);
        stage_issue_n<NWS4>(wp_f + PF::wh(1), PF::kHH, lds_w + kLdsFineW1);
        stage_issue_n<NWS4>(wp_f + PF::wh(2), PF::kHH, lds_w + kLdsFineW2);
    }
}

struct Sampler4Args {
    const float* rays_o;      // [R,3]
    const float* rays_d;      // [R,3]
    const float* t_lin;       // [E] = linspace(0,1,E)
    const float* t_rand;      // [R,E] stratified jitter in [0,1) or nullptr (eval mode)
    float* z;                 // [R,E] out
    float* sdf;               // [R,E] out
    float* far;               // [R] out
    uint32_t R, E;
    float near, bound, far_cap;
    const float* table_c;
    const float* table_f;
    const float* wp_c;
    const float* wp_f;
    float df_c, df_f;
};

// far end of the ray inside the cube [-bound, bound]^3, clamped to far_cap (ray_sampler.py:23-35)
__device__ __forceinline__ float cube_far4(const float (&o)[3], const float (&d)[3], float bound, float far_cap) {
    float nearv = -INFINITY, farv = INFINITY;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float den = d[k] + 1e-15f;
        const float t0 = (-bound - o[k]) / den;
        const float t1 = (bound - o[k]) / den;
        nearv = fmaxf(nearv, t0 < t1 ? t0 : t1);
        farv = fminf(farv, t0 > t1 ? t0 : t1);
    }
    if (farv < nearv) farv = 1e9f;
    return fminf(farv, far_cap);
}

template <int LC, int CC, int NHC, int LF, int CF, int NHF>
__global__ __launch_bounds__(64 * NWS4, NWS4 / 4) void k_sampler4_sdf(Sampler4Args a, GridGeom16 gc, GridGeom16 gf) {
    __shared__ __attribute__((aligned(16))) float lds_w[kLdsWeights];
    __shared__ LevelGeom s_gc[16], s_gf[16];
    load_resident_weights<NHC, NHF>(lds_w, a.wp_c, a.wp_f);
    geom_to_lds(gc, s_gc);
    geom_to_lds(gf, s_gf);
    stage_wait();                                    // vmcnt(0) + workgroup barrier: weights and geometry are in LDS
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const uint64_t total = (uint64_t)a.R * a.E;
    const uint32_t n_tiles = (uint32_t)((total + 15) / 16);
    ResidentGemm<NHC> gemm_c{lds_w + kLdsCoarseW0, lane, a.wp_c};
    ResidentGemm<NHF> gemm_f{lds_w + kLdsFineW0, lane, a.wp_f};
    const uint32_t E = a.E;
    for (uint32_t tile = blockIdx.x * NWS4 + (threadIdx.x >> 6); tile < n_tiles; tile += gridDim.x * NWS4) {
        uint64_t pid = (uint64_t)tile * 16 + j;
        const bool live = pid < total;
        if (!live) pid = total - 1;                  // keep the wave converged for the MFMAs; store is predicated
        const uint32_t ray = (uint32_t)(pid / E);
        const uint32_t i = (uint32_t)(pid - (uint64_t)ray * E);
        float o[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { o[k] = a.rays_o[ray * 3 + k]; d[k] = a.rays_d[ray * 3 + k]; }
        const float farv = cube_far4(o, d, a.bound, a.far_cap);
        const float nearv = a.near;
        // z_lin(i) = near (1 - t_i) + far t_i ; stratified: lower + (upper - lower) * rand   (ray_sampler.py:49-59)
        // every product rounded separately, as the reference's elementwise torch ops do (see mul_rn)
        const float ti = a.t_lin[i];
        float zi = mul_rn(nearv, 1.0f - ti) + mul_rn(farv, ti);
        if (a.t_rand) {
            const float tp = a.t_lin[i + 1 < E ? i + 1 : i], tm = a.t_lin[i > 0 ? i - 1 : 0];
            const float zp = mul_rn(nearv, 1.0f - tp) + mul_rn(farv, tp);
            const float zm = mul_rn(nearv, 1.0f - tm) + mul_rn(farv, tm);
            const float upper = i + 1 < E ? 0.5f * (zp + zi) : zi;
            const float lower = i > 0 ? 0.5f * (zi + zm) : zi;
            zi = lower + mul_rn(upper - lower, a.t_rand[pid]);
        }
        float x[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = o[k] + mul_rn(zi, d[k]);

        float in[QIN];
        pe_slots4(x, q, in);                         // shared by both networks
        grid_slots4<LC, CC>(x, a.df_c, a.table_c, s_gc, q, in);
        float sdf = sdf_only4<NHC>(a.wp_c, q, in, gemm_c);
        grid_slots4<LF, CF>(x, a.df_f, a.table_f, s_gf, q, in);
        sdf += sdf_only4<NHF>(a.wp_f, q, in, gemm_f);
        if (live && q == 0) {
            a.z[pid] = zi;
            a.sdf[pid] = sdf;
            if (i == 0) a.far[ray] = farv;
        }
    }
}

// SDF at explicit points, no gradient (batch inference: mesh extraction grids, plots).
struct SdfPoints4Args {
    const float* points;      // [N,3]
    float* sdf;               // [N]
    uint64_t N;
    const float* table_c;
    const float* table_f;     // nullptr: stage "coarse"
    const float* wp_c;
    const float* wp_f;
    float df_c, df_f;
};

template <int LC, int CC, int NHC, int LF, int CF, int NHF>
__global__ __launch_bounds__(64 * NWS4, NWS4 / 4) void k_sdf4_points(SdfPoints4Args a, GridGeom16 gc, GridGeom16 gf) {
    __shared__ __attribute__((aligned(16))) float lds_w[kLdsWeights];
    __shared__ LevelGeom s_gc[16], s_gf[16];
    load_resident_weights<NHC, NHF>(lds_w, a.wp_c, a.table_f ? a.wp_f : nullptr);
    geom_to_lds(gc, s_gc);
    geom_to_lds(gf, s_gf);
    stage_wait();
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const uint64_t n_tiles = (a.N + 15) / 16;
    ResidentGemm<NHC> gemm_c{lds_w + kLdsCoarseW0, lane, a.wp_c};
    ResidentGemm<NHF> gemm_f{lds_w + kLdsFineW0, lane, a.wp_f};
    for (uint64_t tile = (uint64_t)blockIdx.x * NWS4 + (threadIdx.x >> 6); tile < n_tiles; tile += (uint64_t)gridDim.x * NWS4) {
        uint64_t pid = tile * 16 + j;
        const bool live = pid < a.N;
        if (!live) pid = a.N - 1;
        float x[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = a.points[pid * 3 + k];
        float in[QIN];
        pe_slots4(x, q, in);
        grid_slots4<LC, CC>(x, a.df_c, a.table_c, s_gc, q, in);
        float sdf = sdf_only4<NHC>(a.wp_c, q, in, gemm_c);
        if (a.table_f) {                             // uniform branch
            grid_slots4<LF, CF>(x, a.df_f, a.table_f, s_gf, q, in);
            sdf += sdf_only4<NHF>(a.wp_f, q, in, gemm_f);
        }
        if (live && q == 0) a.sdf[pid] = sdf;
    }
}

static int persistent_blocks() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            n = cus;
        else
            n = 256;
    }
    return n;
}

}  // namespace nsa

#include "quad_entries.hpp"

// Internal entry points (not in the public header): render_sampler.hip forwards here when nsa_grid_t.tile == 16.
extern "C" {

int NSA_ENTRY(nsa_sampler4_sdf)(const float* rays_o, const float* rays_d, uint32_t R, uint32_t E, const float* t_lin,
                                const float* t_rand, float near, float bound, float far_cap, const nsa_grid_t* coarse,
                                const nsa_grid_t* fine, const float* packed_coarse, const float* packed_fine, float* z, float* sdf,
                                float* far, nsa_stream_t stream) {
    using namespace nsa;
    GridGeom16 gc, gf;
    if (int rc = make_grid_geom16(coarse->offsets_host, coarse->L, coarse->S, coarse->H, &gc, coarse->C)) return rc;
    if (int rc = make_grid_geom16(fine->offsets_host, fine->L, fine->S, fine->H, &gf, fine->C)) return rc;
    Sampler4Args a{rays_o, rays_d, t_lin, t_rand, z, sdf, far, R, E, near, bound, far_cap,
                   coarse->table, fine->table, packed_coarse, packed_fine, coarse->divide_factor, fine->divide_factor};
    const uint64_t total = (uint64_t)R * E;
    const uint64_t tiles = (total + 15) / 16;
    uint64_t blocks = (tiles + NWS4 - 1) / NWS4;
    if (blocks > (uint64_t)persistent_blocks()) blocks = persistent_blocks();
    launch_begin();
    hipLaunchKernelGGL((k_sampler4_sdf<4, 8, 1, 8, 4, 3>), dim3((uint32_t)blocks), dim3(64 * NWS4), 0, (hipStream_t)stream, a, gc, gf);
    return launch_end();
}

int NSA_ENTRY(nsa_sdf4_points)(const float* points, uint64_t N, const nsa_grid_t* coarse, const nsa_grid_t* fine,
                               const float* packed_coarse, const float* packed_fine, float* sdf, nsa_stream_t stream) {
    using namespace nsa;
    GridGeom16 gc, gf{};
    if (int rc = make_grid_geom16(coarse->offsets_host, coarse->L, coarse->S, coarse->H, &gc, coarse->C)) return rc;
    if (fine) if (int rc = make_grid_geom16(fine->offsets_host, fine->L, fine->S, fine->H, &gf, fine->C)) return rc;
    SdfPoints4Args a{points, sdf, N, coarse->table, fine ? fine->table : nullptr, packed_coarse, fine ? packed_fine : nullptr,
                     coarse->divide_factor, fine ? fine->divide_factor : 1.0f};
    const uint64_t tiles = (N + 15) / 16;
    uint64_t blocks = (tiles + NWS4 - 1) / NWS4;
    if (blocks > (uint64_t)persistent_blocks()) blocks = persistent_blocks();
    launch_begin();
    hipLaunchKernelGGL((k_sdf4_points<4, 8, 1, 8, 4, 3>), dim3((uint32_t)blocks), dim3(64 * NWS4), 0, (hipStream_t)stream, a, gc, gf);
    return launch_end();
}

}  // extern "C"
