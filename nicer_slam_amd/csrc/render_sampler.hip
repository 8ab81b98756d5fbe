// render_sampler.hip -- the hierarchical ray sampler of the render core.
// Reference: UniformSampler.get_z_vals / near_far_from_cube (code/model/ray_sampler.py:23-61),
//            ImportantSampler.get_z_vals (code/model/ray_sampler.py:90-166),
//            GridPredefineDensity (code/model/density.py:37-67).
//
// k_sampler_sdf   one lane-pair per (ray, coarse sample): builds the stratified z, the point, both grid encodings,
//                 the positional encoding and evaluates coarse+fine SDF MLPs on the matrix cores (fp32-faithful
//                 split GEMM, mlp_common.hpp) with all activations in registers
//                 (the reference's redundant second coarse evaluation, base_networks.py:31, is not repeated).
// k_sample_rays   one workgroup per ray: SDF -> Laplace density (beta from the visit counter) -> alpha/transmittance
//                 weights via a workgroup scan -> pdf/cdf in LDS -> inverse-CDF samples by binary search ->
//                 merge with near/far/extras -> rank sort out of LDS.
#include "sampler_common.hpp"
#include "draw_common.hpp"

namespace nsa {

// T = point tiles (of 32 points) per wave.  T = 1: the round-1 form, 167 registers, three waves per SIMD.  T = 2: one weight
// fragment stream serves both tiles (sdf_only_tiles), two waves per SIMD.
#ifdef NSA_X_TS
static __device__ unsigned long long* g_ts_s = nullptr;
#define STS_BEGIN const unsigned long long ts_start = ts_now(); unsigned long long ts_prev = ts_start; \
    if ((threadIdx.x & 63) == 0) for (int i = 0; i < 16; ++i) nsa_ts_lds[threadIdx.x >> 6][i] = 0;
#define STS_MARK(slot) { const unsigned long long t_ = ts_now(); ts_add(slot, t_ - ts_prev); ts_prev = t_; }
#define STS_END { ts_add(15, ts_now() - ts_start); if (g_ts_s && (threadIdx.x & 63) == 0) { \
    unsigned long long* o_ = g_ts_s + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16; \
    for (int i = 0; i < 16; ++i) o_[i] = nsa_ts_lds[threadIdx.x >> 6][i]; \
    o_[0] = ts_start; o_[1] = ts_now(); /* absolute: slot timeline (tools/slot_timeline.py) */ \
    o_[2] = (unsigned long long)__builtin_amdgcn_s_getreg(0xF804) | ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32); } }
#else
#define STS_BEGIN
#define STS_MARK(slot)
#define STS_END
#endif
// waves per SIMD the register allocation aims at: two tiles per wave need the whole file (2); one tile per wave fits 3 with the
// operand split's fragments and 4 (128 registers) in the bf16-operand build, whose weight fragments are a third of the size
#ifdef NSA_OCC_SAMPLER_T1
constexpr int kOccSamplerT1 = NSA_OCC_SAMPLER_T1;
#else
constexpr int kOccSamplerT1 = kPieces == 1 ? 4 : 3;
#endif
template <int LC, int CC, int NHC, int LF, int CF, int NHF, int T>
__global__ __launch_bounds__(256, T == 1 ? kOccSamplerT1 : NSA_OCC_SAMPLER) void k_sampler_sdf(SamplerArgs a, GridGeom16 gc, GridGeom16 gf) {
    STS_BEGIN
    desync_simd_partners();
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t total = (uint64_t)a.R * a.E;
    // z and far are final as soon as the point exists: they are stored at once, so that only the point index stays live across the
    // two networks (round 5: the two-tile form had 4 spilled registers -- zi / farv / ray / idx of both tiles were carried to the end)
    uint32_t pid[T];          // pid < 2^32: the entry point refuses larger launches for this kernel (64-bit indexing is ~100 VALU per tile)
    bool live[T];
    float x[T][3];
    float in[T][SDF_IN_STEPS];
    RayOfTile rt;
    uint32_t prev_ray = 0xFFFFFFFFu;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const uint64_t p64 = ((uint64_t)wave * T + t) * 32 + (lane & 31);
        live[t] = p64 < total;
        pid[t] = live[t] ? (uint32_t)p64 : (uint32_t)(total - 1);       // keep the wave converged for the MFMAs; stores are predicated
        const uint32_t ray = pid[t] / a.E;
        const uint32_t idx = pid[t] - ray * a.E;
        // the ray's origin, direction and far end (six divisions) are shared by the wave's tiles whenever they lie on one ray --
        // always at the shipped E = 640 = 10 x 64
        if (t == 0 || !__all(ray == prev_ray)) ray_of_tile(a, ray, rt);
        prev_ray = ray;
        float zi, farv;
        sampler_point(a, pid[t], rt, idx, x[t], zi, farv);
        if (live[t] && h == 0) {
            a.z[pid[t]] = zi;
            if (idx == 0) a.far[ray] = farv;
        }
        STS_MARK(4 + 3 * 0)
        pe_slots(x[t], h, in[t]);               // shared by both networks
        STS_MARK(5)
        grid_slots<LC, CC, true>(x[t], a.df_c, a.table_c, gc, h, in[t]);
        STS_MARK(6)
    }
    float sdf[T], sdf_f[T];
    sdf_only_tiles<NHC, T>(a.wp_c, lane, h, in, sdf);
    STS_MARK(7)
#pragma unroll
    for (int t = 0; t < T; ++t) grid_slots<LF, CF, true>(x[t], a.df_f, a.table_f, gf, h, in[t]);
    STS_MARK(8)
    sdf_only_tiles<NHF, T>(a.wp_f, lane, h, in, sdf_f);
    STS_MARK(9)
#pragma unroll
    for (int t = 0; t < T; ++t)
        if (live[t] && h == 0) a.sdf[pid[t]] = sdf[t] + sdf_f[t];
    STS_MARK(10)
    STS_END
}


// SDF at explicit points, no gradient (batch inference: mesh extraction grids, plots).
// Reference: ImplicitNetworkGrid_COMBINE.get_sdf_vals (code/model/base_networks.py:25-35).  table_f == nullptr: stage "coarse".
struct SdfPointsArgs {
    const float* points;      // [N,3]
    float* sdf;               // [N]
    uint64_t N;
    const float* table_c;
    const float* table_f;
    const float* wp_c;
    const float* wp_f;
    float df_c, df_f;
};

template <int LC, int CC, int NHC, int LF, int CF, int NHF>
__global__ __launch_bounds__(256, 2) void k_sdf_points(SdfPointsArgs a, GridGeom16 gc, GridGeom16 gf) {
    desync_simd_partners();
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint64_t pid = wave * 32 + (lane & 31);
    const bool live = pid < a.N;
    if (!live) pid = a.N - 1;
    float x[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = a.points[pid * 3 + k];
    float in[SDF_IN_STEPS];
    pe_slots(x, h, in);
    grid_slots<LC, CC, true>(x, a.df_c, a.table_c, gc, h, in);
    float sdf = sdf_only<NHC>(a.wp_c, lane, h, in);
    if (a.table_f) {                            // uniform branch
        grid_slots<LF, CF, true>(x, a.df_f, a.table_f, gf, h, in);
        sdf += sdf_only<NHF>(a.wp_f, lane, h, in);
    }
    if (live && h == 0) a.sdf[pid] = sdf;
}

// Per-iteration integer draws of the sampler from ONE buffer of uniforms (training mode):
//   extra_idx = first n_extra entries of a random permutation of 0..E-1  (torch.randperm(E)[:n], ray_sampler.py:148)
//               = indices of the n_extra smallest of E i.i.d. uniform keys, in key order (rank by counting in LDS);
//   eik_idx[r] = floor(u * S) in 0..S-1                                  (torch.randint(S, (R,)), ray_sampler.py:158).
// Replaces a rand + argsort (radix sort, arange, fills, casts) + randint chain of ~10 launches.
__global__ __launch_bounds__(256) void k_draw_picks(const float* __restrict__ u, uint32_t E, uint32_t n_extra, uint32_t R,
                                                    uint32_t S, int32_t* __restrict__ extra_idx, int32_t* __restrict__ eik_idx) {
    // 64 keys per workgroup (E/64 workgroups on different CUs: the E^2 comparisons are what costs); every workgroup holds
    // all E keys in LDS; its four waves each count one quarter of the comparison range with 16-byte broadcast reads
    __shared__ __attribute__((aligned(16))) float key[1024];
    __shared__ uint32_t part[4][64];
    const uint32_t lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < 1024; i += 256) key[i] = i < E ? u[i] : 2.0f;   // pad keys never rank below a real key
    __syncthreads();
    const uint32_t t = blockIdx.x * 64 + lane;
    {
        const float mine = key[t < 1024 ? t : 1023];
        uint32_t rank = 0;
        const float4* k4 = reinterpret_cast<const float4*>(key);
        const uint32_t n4 = (E + 3) / 4, per = (n4 + 3) / 4;
        const uint32_t lo = q * per, hi = lo + per < n4 ? lo + per : n4;
#pragma unroll 8
        for (uint32_t j4 = lo; j4 < hi; ++j4) {
            const float4 o = k4[j4];
            const uint32_t j = 4 * j4;
            rank += (o.x < mine || (o.x == mine && j < t)) ? 1u : 0u;
            rank += (o.y < mine || (o.y == mine && j + 1 < t)) ? 1u : 0u;
            rank += (o.z < mine || (o.z == mine && j + 2 < t)) ? 1u : 0u;
            rank += (o.w < mine || (o.w == mine && j + 3 < t)) ? 1u : 0u;
        }
        part[q][lane] = rank;
    }
    __syncthreads();
    if (q == 0 && extra_idx && t < E) {
        const uint32_t rank = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
        if (rank < n_extra) extra_idx[rank] = (int32_t)t;
    }
    if (eik_idx)
        for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < R; r += gridDim.x * 256) {
            const uint32_t v = (uint32_t)(u[E + r] * (float)S);
            eik_idx[r] = (int32_t)(v < S ? v : S - 1);
        }
}

// ---- all random draws of one sampler call in ONE launch: draw_common.hpp (shared with the tracker's head launch) ----------------
__global__ __launch_bounds__(256) void k_draw(DrawArgs a) { draw_block(a, blockIdx.x, gridDim.x); }

// ------------------------------------------------------------------------------------------------ per-ray stage
struct RaySampleArgs {
    const float* rays_o;
    const float* rays_d;
    const float* z;         // [R,E] coarse samples
    const float* sdf;       // [R,E]
    const float* far;       // [R]
    const float* voxels;    // [res^3] visit counter
    const float* u_lin;     // [N] = linspace(0,1,N)
    const int32_t* extra_idx;   // [n_extra] indices into the E coarse samples
    const int32_t* eik_idx;     // [R] or nullptr
    float* z_vals;          // [R,S] out, S = N + 2 + n_extra
    float* z_eik;           // [R] out (may be nullptr)
    uint32_t R, E, N, n_extra, voxel_res;
    float near;
};

__device__ __forceinline__ float beta_of(const float* __restrict__ voxels, uint32_t res, const float (&x)[3]) {
    // density.py:41-60
    const bool outside = fabsf(x[0]) > 0.99f || fabsf(x[1]) > 0.99f || fabsf(x[2]) > 0.99f;
    float count = 0.0f;
    if (!outside) {
        int idx[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int v = (int)((x[k] + 1.0f) / 2.0f * (float)res);   // .long() truncation
            idx[k] = v < 0 ? 0 : (v >= (int)res ? (int)res - 1 : v);
        }
        count = voxels[((uint32_t)idx[0] * res + (uint32_t)idx[1]) * res + (uint32_t)idx[2]];     // res <= 1024 (entry check)
    }
    return 0.01207724805f * expf(-0.0116544676f * 0.0001f * count * 5.37538f) + 0.0023639156f;
}

__device__ __forceinline__ float laplace_density(float sdf, float beta) {
    // alpha * (0.5 + 0.5 * sign(s) * expm1(-|s| / beta)), alpha = 1/beta   (density.py:37-39)
    const float sg = sdf > 0.0f ? 1.0f : (sdf < 0.0f ? -1.0f : 0.0f);
    return (1.0f / beta) * (0.5f + 0.5f * sg * expm1f(-fabsf(sdf) / beta));
}

constexpr int MAX_S = 256;    // final samples per ray supported by the sort buffer

// Exclusive prefix of v over the 256 threads of the workgroup (thread order) and the workgroup total: wave-shuffle scan, the
// four wave totals through LDS, added in wave order.  `red`: 4 floats of LDS, free again on return.
__device__ __forceinline__ float block_excl_scan(float v, int lane, int wv, float* red, float& total) {
    float incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float n = __shfl_up(incl, off);
        if (lane >= off) incl += n;
    }
    if (lane == 63) red[wv] = incl;
    // exclusive = inclusive of the previous lane (NOT incl - v: the last sample's free energy is ~1e10 * sigma and the
    // subtraction would cancel the whole prefix)
    const float prev = __shfl_up(incl, 1);
    __syncthreads();
    float base = 0.0f;
    for (int w = 0; w < wv; ++w) base += red[w];
    total = ((red[0] + red[1]) + red[2]) + red[3];
    __syncthreads();
    return lane == 0 ? base : base + prev;
}

// One 256-thread workgroup per ray (round 3; rounds 1-2 ran one WAVE per ray: at 1024 rays that is a single wave per SIMD, every
// load latency and every dependent instruction exposed -- 21 us of which 19 were on the iteration's critical path).  LDS per ray:
// pdf[E], cdf[E], z[E], keys[MAX_S + 4].  Four workgroups share a CU.
__global__ __launch_bounds__(256) void k_sample_rays(RaySampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t E = a.E, N = a.N;
    const uint32_t S = N + 2 + a.n_extra;
    const uint32_t ray = blockIdx.x;
    const uint32_t E4 = (E + 3) & ~3u;
    float* pdf = smem;
    float* cdf = pdf + E;
    float* zb = cdf + E;
    float* sb = smem + 3 * (size_t)E4;                  // 16-byte aligned: the sort reads it four keys at a time
    STS_BEGIN

    float o[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[k] = a.rays_o[ray * 3 + k]; d[k] = a.rays_d[ray * 3 + k]; }
    const float* zr = a.z + (size_t)ray * E;
    const float* sr = a.sdf + (size_t)ray * E;

    // free energy sigma_i * delta_i (last delta = 1e10)                       ray_sampler.py:105-108
    // thread-strided (coalesced loads; a thread's iterations -- each a z load, an sdf load and a dependent visit-counter
    // gather -- are all in flight at once); the scans below are thread-blocked and read what this phase left in LDS.
#pragma unroll 3
    for (uint32_t i = tid; i < E; i += 256) {
        const float zi = zr[i];
        const float zn = i + 1 < E ? zr[i + 1] : 0.0f;
        float x[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = o[c] + mul_rn(zi, d[c]);
        const float sigma = laplace_density(sr[i], beta_of(a.voxels, a.voxel_res, x));
        zb[i] = zi;
        pdf[i] = (i + 1 < E ? zn - zi : 1e10f) * sigma;
    }
    __syncthreads();
    STS_MARK(1)
    const uint32_t per = (E + 255) / 256;
    const uint32_t i0 = tid * per;
    float esum = 0.0f;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t i = i0 + k;
        if (i < E) esum += pdf[i];
    }
    float tot;
    float run = block_excl_scan(esum, lane, wv, red, tot);
    STS_MARK(2)
    // w_i = (1 - exp(-E_i)) exp(-sum_{j<i} E_j);  pdf_i = w_i + 1e-5 for i < E-1      :109-117
    float psum = 0.0f;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t i = i0 + k;
        if (i < E) {
            const float en = pdf[i];
            const float w = (1.0f - expf(-en)) * expf(-run);
            run += en;
            const float p = i + 1 < E ? w + 1e-5f : 0.0f;
            pdf[i] = p;
            psum += p;
        }
    }
    STS_MARK(3)
    float ptot;
    (void)block_excl_scan(psum, lane, wv, red, ptot);
    STS_MARK(4)
    // cdf = [0, cumsum(pdf / sum)]                                                     :118-121
    float nsum = 0.0f;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t i = i0 + k;
        if (i + 1 < E) {
            const float q = pdf[i] / ptot;
            pdf[i] = q;                                  // (thread-private elements: no barrier needed before the re-read)
            nsum += q;
        }
    }
    float ntot;
    float nrun = block_excl_scan(nsum, lane, wv, red, ntot);
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t i = i0 + k;
        if (i + 1 < E) {
            nrun += pdf[i];
            cdf[i + 1] = nrun;
        }
    }
    if (tid == 0) cdf[0] = 0.0f;
    __syncthreads();
    STS_MARK(5)

    // inverse CDF at u_j = linspace(0,1,N)_j: searchsorted(right=True)                 :124-139
    for (uint32_t j = tid; j < N; j += 256) {
        const float u = a.u_lin[j];
        uint32_t lo = 0, hi = E;                     // first index with cdf > u
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const uint32_t below = lo > 0 ? lo - 1 : 0;
        const uint32_t above = lo < E - 1 ? lo : E - 1;
        const float c0 = cdf[below], c1 = cdf[above];
        const float b0 = zb[below], b1 = zb[above];
        float den = c1 - c0;
        if (den < 1e-5f) den = 1.0f;
        sb[j] = b0 + mul_rn((u - c0) / den, b1 - b0);
    }
    // extras: near, far, n_extra of the coarse samples                                  :146-153
    const uint32_t S4 = (S + 3) & ~3u;
    for (uint32_t j = N + tid; j < S4; j += 256) {
        float v = INFINITY;                               // padding of the last group of four keys
        if (j == N) v = a.near;
        else if (j == N + 1) v = a.far[ray];
        else if (j < S) {                                 // caller-supplied pick: clamped to the ray's E coarse samples
            const uint32_t e = (uint32_t)a.extra_idx[j - N - 2];
            v = zb[e < E ? e : E - 1];
        }
        sb[j] = v;
    }
    __syncthreads();
    STS_MARK(6)
    // sort (:155) by rank: key t goes to position #{j : key_j < key_t} + #{j < t : key_j == key_t}.  Every thread reads the
    // same four keys per step (an LDS broadcast), no exchanges, no barriers -- S <= 256 keys cost S/4 steps.
    uint32_t e_pick = S;
    if (a.z_eik) {
        const uint32_t e = (uint32_t)a.eik_idx[ray];
        e_pick = e < S ? e : S - 1;
    }
    // S <= 128 (the shipped 98): two threads per key, each counts over half of the keys; the halves meet in LDS (the unused
    // upper half of the key buffer).
    const bool two = S <= 128;
    const uint32_t t0 = two ? (uint32_t)tid & 127u : (uint32_t)tid;
    const uint32_t half = two ? (uint32_t)tid >> 7 : 0u;
    const uint32_t jmid = two ? ((S4 >> 1) + 3) & ~3u : S4;
    const uint32_t jlo = half ? jmid : 0u, jhi = half ? S4 : jmid;
    uint32_t* cnt = reinterpret_cast<uint32_t*>(sb + S4);       // two: S4 + 128 <= MAX_S
    const float key = t0 < S ? sb[t0] : 0.0f;
    uint32_t rank = 0;
    if (t0 < S) {
        for (uint32_t j = jlo; j < jhi; j += 4) {
            const float4 v = *reinterpret_cast<const float4*>(sb + j);
            rank += (v.x < key || (v.x == key && j + 0 < t0)) ? 1u : 0u;
            rank += (v.y < key || (v.y == key && j + 1 < t0)) ? 1u : 0u;
            rank += (v.z < key || (v.z == key && j + 2 < t0)) ? 1u : 0u;
            rank += (v.w < key || (v.w == key && j + 3 < t0)) ? 1u : 0u;
        }
        if (half) cnt[t0] = rank;
    }
    if (two) __syncthreads();
    if (t0 < S && !half) {
        if (two) rank += cnt[t0];
        a.z_vals[(size_t)ray * S + rank] = key;
        if (rank == e_pick) a.z_eik[ray] = key;
    }
    STS_MARK(7)
    STS_END
}

}  // namespace nsa

// Entry-point naming: this file is compiled twice -- as is (fp32-faithful GEMMs) and through *_bf16.hip with
// NSA_PIECES = 1, `nsa` renamed and every entry point suffixed _bf16; the fp32 entry points forward to those when
// nsa_grid_t.precision == 1.
#ifndef NSA_ENTRY
#define NSA_ENTRY(x) x
#endif
#include "bf16_entries.hpp"
#include "quad_entries.hpp"
// The wave-specialised samplers (tile codes 96 / 97; render_sampler_ws.hip, render_sampler_sys.hip) are experiment kernels: compiled
// only into a tagged side-by-side build (build.py, NSA_X_WS=1 -> NSA_X_WS_SAMPLERS); the product library refuses the two codes.
#ifdef NSA_X_WS_SAMPLERS
extern "C" int nsa_sampler_sys_sdf(const float* rays_o, const float* rays_d, uint32_t R, uint32_t E, const float* t_lin,
                                   const float* t_rand, float near, float bound, float far_cap, const nsa_grid_t* coarse,
                                   const nsa_grid_t* fine, const float* packed_coarse, const float* packed_fine, float* z, float* sdf,
                                   float* far, nsa_stream_t stream);
extern "C" int nsa_sampler_ws_sdf(const float* rays_o, const float* rays_d, uint32_t R, uint32_t E, const float* t_lin,
                                  const float* t_rand, float near, float bound, float far_cap, const nsa_grid_t* coarse,
                                  const nsa_grid_t* fine, const float* packed_coarse, const float* packed_fine, float* z, float* sdf,
                                  float* far, nsa_stream_t stream);
#endif

extern "C" {

int NSA_ENTRY(nsa_sampler_sdf)(const float* rays_o, const float* rays_d, uint32_t R, uint32_t E, const float* t_lin,
                    const float* t_rand, float near, float bound, float far_cap, const nsa_grid_t* coarse,
                    const nsa_grid_t* fine, const float* packed_coarse, const float* packed_fine, float* z, float* sdf,
                    float* far, nsa_stream_t stream) {
#if NSA_PIECES != 1
    if (coarse && coarse->precision == 1) return nsa_sampler_sdf_bf16(rays_o, rays_d, R, E, t_lin, t_rand, near, bound, far_cap, coarse, fine, packed_coarse, packed_fine, z, sdf, far, stream);      // bf16-operand kernels (csrc/*_bf16.hip)
#endif
    using namespace nsa;
    if (R == 0 || E == 0) return NSA_OK;
    if (!rays_o || !rays_d || !t_lin || !coarse || !fine || !packed_coarse || !packed_fine || !z || !sdf || !far)
        return NSA_EBADARG;
    if (!(coarse->L == 4 && coarse->C == 8 && coarse->n_hidden == 1 && fine->L == 8 && fine->C == 4 && fine->n_hidden == 3))
        return NSA_EUNSUPPORTED_NET;
#if NSA_PIECES == 3 && defined(NSA_X_WS_SAMPLERS)
    if (coarse->tile == 97 || fine->tile == 97) {         // systolic wave-specialised form (render_sampler_sys.hip)
        if (coarse->tile != fine->tile) return NSA_EBADARG;
        return nsa_sampler_sys_sdf(rays_o, rays_d, R, E, t_lin, t_rand, near, bound, far_cap, coarse, fine, packed_coarse,
                                   packed_fine, z, sdf, far, stream);
    }
    if (coarse->tile == 96 || fine->tile == 96) {         // wave-specialised form (render_sampler_ws.hip)
        if (coarse->tile != fine->tile) return NSA_EBADARG;
        return nsa_sampler_ws_sdf(rays_o, rays_d, R, E, t_lin, t_rand, near, bound, far_cap, coarse, fine, packed_coarse,
                                  packed_fine, z, sdf, far, stream);
    }
#else
    if (coarse->tile == 96 || coarse->tile == 97 || fine->tile == 96 || fine->tile == 97) return NSA_EUNSUPPORTED_NET;
#endif
    if (coarse->tile == 16 || fine->tile == 16) {
        if (coarse->tile != fine->tile) return NSA_EBADARG;
        return NSA_ENTRY(nsa_sampler4_sdf)(rays_o, rays_d, R, E, t_lin, t_rand, near, bound, far_cap, coarse, fine, packed_coarse,
                                           packed_fine, z, sdf, far, stream);
    }
    GridGeom16 gc, gf;
    if (int rc = make_grid_geom16(coarse->offsets_host, coarse->L, coarse->S, coarse->H, &gc, coarse->C)) return rc;
    if (int rc = make_grid_geom16(fine->offsets_host, fine->L, fine->S, fine->H, &gf, fine->C)) return rc;
    SamplerArgs a{rays_o, rays_d, t_lin, t_rand, z, sdf, far, R, E, near, bound, far_cap,
                  coarse->table, fine->table, packed_coarse, packed_fine, coarse->divide_factor, fine->divide_factor};
    const uint64_t total = (uint64_t)R * E;
    if (total > 0xFFFFFFFFull) return NSA_EBADARG;        // 32-bit point indices in k_sampler_sdf (callers chunk far below this)
    const bool two = coarse->tile == 64;                  // 64: 32-point tiling, two tiles per wave
    if (two != (fine->tile == 64)) return NSA_EBADARG;
    const uint32_t waves = (uint32_t)((total + (two ? 63 : 31)) / (two ? 64 : 32));
    const uint32_t blocks = (waves + 3) / 4;
    launch_begin();
    if (two) hipLaunchKernelGGL((k_sampler_sdf<4, 8, 1, 8, 4, 3, 2>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, gc, gf);
    else     hipLaunchKernelGGL((k_sampler_sdf<4, 8, 1, 8, 4, 3, 1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, gc, gf);
    return launch_end();
}

int NSA_ENTRY(nsa_sdf_points)(const float* points, uint64_t N, const nsa_grid_t* coarse, const nsa_grid_t* fine,
                   const float* packed_coarse, const float* packed_fine, float* sdf, nsa_stream_t stream) {
#if NSA_PIECES != 1
    if (coarse && coarse->precision == 1) return nsa_sdf_points_bf16(points, N, coarse, fine, packed_coarse, packed_fine, sdf, stream);      // bf16-operand kernels (csrc/*_bf16.hip)
#endif
    using namespace nsa;
    if (N == 0) return NSA_OK;
    if (!points || !coarse || !packed_coarse || !sdf || (fine && !packed_fine)) return NSA_EBADARG;
    if (!(coarse->L == 4 && coarse->C == 8 && coarse->n_hidden == 1)) return NSA_EUNSUPPORTED_NET;
    if (fine && !(fine->L == 8 && fine->C == 4 && fine->n_hidden == 3)) return NSA_EUNSUPPORTED_NET;
    if (coarse->tile == 16) {
        if (fine && fine->tile != 16) return NSA_EBADARG;
        return NSA_ENTRY(nsa_sdf4_points)(points, N, coarse, fine, packed_coarse, packed_fine, sdf, stream);
    }
    GridGeom16 gc, gf{};
    if (int rc = make_grid_geom16(coarse->offsets_host, coarse->L, coarse->S, coarse->H, &gc, coarse->C)) return rc;
    if (fine) if (int rc = make_grid_geom16(fine->offsets_host, fine->L, fine->S, fine->H, &gf, fine->C)) return rc;
    SdfPointsArgs a{points, sdf, N, coarse->table, fine ? fine->table : nullptr, packed_coarse, fine ? packed_fine : nullptr,
                    coarse->divide_factor, fine ? fine->divide_factor : 1.0f};
    const uint64_t waves = (N + 31) / 32;
    const uint64_t blocks = (waves + 3) / 4;
    if (blocks > 0x7FFFFFFFull) return NSA_EBADARG;
    launch_begin();
    hipLaunchKernelGGL((k_sdf_points<4, 8, 1, 8, 4, 3>), dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, a, gc, gf);
    return launch_end();
}

#ifdef NSA_X_TS
int NSA_ENTRY(nsa_debug_set_ts_sampler)(unsigned long long* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(nsa::g_ts_s), &p, sizeof(p)) == hipSuccess ? 0 : 3;
}
#endif

int NSA_ENTRY(nsa_draw_picks)(const float* u, uint32_t E, uint32_t n_extra, uint32_t R, uint32_t S, int32_t* extra_idx,
                   int32_t* eik_idx, nsa_stream_t stream) {
    using namespace nsa;
    if (!u || E == 0 || E > 1024 || n_extra > E || (eik_idx && S == 0)) return NSA_EBADARG;
    launch_begin();
    hipLaunchKernelGGL(k_draw_picks, dim3((E + 63) / 64), dim3(256), 0, (hipStream_t)stream, u, E, n_extra, R, S, extra_idx, eik_idx);
    return launch_end();
}

int NSA_ENTRY(nsa_draw)(uint64_t* state, uint64_t n_rand, float* t_rand, uint32_t E, uint32_t n_extra, uint32_t R, uint32_t S,
                        int32_t* extra_idx, int32_t* eik_idx, nsa_stream_t stream) {
    using namespace nsa;
    if (!draw_args_ok(state, n_rand, t_rand, E, n_extra, R, S, extra_idx, eik_idx)) return NSA_EBADARG;
    DrawArgs a{};
    const uint32_t blocks = draw_launch_shape(a, reinterpret_cast<unsigned long long*>(state), n_rand, t_rand, E, n_extra, R, S, extra_idx,
                                              eik_idx);
    if (blocks == 0) return NSA_OK;
    launch_begin();
    hipLaunchKernelGGL(k_draw, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return launch_end();
}

int NSA_ENTRY(nsa_sample_rays)(const float* rays_o, const float* rays_d, const float* z, const float* sdf, const float* far,
                    const float* voxels, uint32_t voxel_res, uint32_t R, uint32_t E, uint32_t N, const float* u_lin,
                    const int32_t* extra_idx, uint32_t n_extra, float near, const int32_t* eik_idx, float* z_vals,
                    float* z_eik, nsa_stream_t stream) {
    using namespace nsa;
    if (R == 0) return NSA_OK;
    if (!rays_o || !rays_d || !z || !sdf || !far || !voxels || !u_lin || !z_vals || (n_extra && !extra_idx) ||
        (z_eik && !eik_idx))
        return NSA_EBADARG;
    const uint32_t S = N + 2 + n_extra;
    if (S > MAX_S || E < 2 || voxel_res == 0 || voxel_res > 1024 || (3 * E + MAX_S) * 4 * 4 > 160 * 1024) return NSA_EBADARG;
    RaySampleArgs a{rays_o, rays_d, z, sdf, far, voxels, u_lin, extra_idx, eik_idx, z_vals, z_eik, R, E, N, n_extra,
                    voxel_res, near};
    const size_t lds = (size_t)(3 * ((E + 3) & ~3u) + MAX_S + 4) * sizeof(float);
    launch_begin();
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)k_sample_rays, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_sample_rays, dim3(R), dim3(256), lds, (hipStream_t)stream, a);
    return launch_end();
}

}  // extern "C"
