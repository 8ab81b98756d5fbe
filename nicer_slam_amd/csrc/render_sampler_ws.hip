// render_sampler_ws.hip -- the sampler's SDF pass (reference code/model/ray_sampler.py:90-112,
// code/model/base_networks.py:155-228) as a WAVE-SPECIALISED kernel: vector work and matrix work run in different waves of one
// workgroup, so the two pipes of a SIMD overlap by construction instead of by the compiler's instruction schedule.
//
// Why.  In k_sampler_sdf (render_sampler.hip) every wave runs the whole per-point program; its time is VALU time + MFMA time +
// exposed latencies (PMC, profiles/r03_pmc_per_kernel.csv: matrix pipe 31 % busy, 64.6 M VALU instructions at ~2.5 cycles of a
// 443 k-cycle launch = 37 %, only half of the matrix-busy cycles with a vector instruction beside them).  A wave issues in
// order: while it waits for a gather, an MFMA result or a transcendental, neither pipe gets work from it, and two such waves per
// SIMD drift into the same phase.  tools/micro/mfma_coissue.hip: an MFMA-only wave and a VALU-only wave on one SIMD overlap 74-96 %.
//
// Roles (one persistent 16-wave workgroup per CU, 128 registers per lane):
//   V waves (8; two per SIMD)  the per-point program WITHOUT its GEMMs: stratified z, point, positional encoding, both corner
//        gathers, softplus, the exact 3-way bf16 split of every GEMM input (mlp_common.hpp::split8).  A V wave owns one 32-point
//        tile at a time and a 15 KiB LDS slot: it writes the B fragments of a layer there (the register image the MFMA wants:
//        [slot group][piece][lane] x 16 B, no transposition), raises `ready`, and picks the accumulators up from the same slot.
//   M waves (8; two per SIMD)  one (layer, 32-feature output tile) each: coarse layer 0, fine layers 0 / 1 / 2, two output tiles.
//        WEIGHT-STATIONARY: the wave's 32 x K block of split weights (K/16 x 3 pieces x 16 B per lane = 48-60 registers) is
//        loaded once per launch and stays in registers -- no weight-fragment traffic at all (k_sampler_sdf streams 108 KB of
//        fragments per pair of tiles through the L1).  The loop: find a client whose B fragments are ready, ds_read_b128 them,
//        issue the six-product MFMA groups in the same order as mlp_common.hpp::mma_group (so the result is BIT-IDENTICAL to
//        k_sampler_sdf), write the 32 x 32 accumulator tile back into the client's slot, raise `done`.
// Hand-off = LDS flags with workgroup-scope release / acquire, no s_barrier after start-up: every wave free-runs, a waiting wave
// sleeps (s_sleep) and leaves its SIMD's issue slots to the others.  The two M waves of a layer read the same B fragments and write
// their accumulator halves over them: each announces "my reads are done" (`bread`) and waits for the other's before writing.
//
// LDS: 8 slots x 15 KiB + bias table 1 KiB + flags = 121.3 KiB of 160.  LDS traffic per 32-point tile: 54 KB written + 108 KB
// read (B), 32 + 32 KB (accumulators) = 226 KB = 1766 clocks of the CU's 128 B/clk.
#include "sampler_common.hpp"

namespace nsa {

constexpr int WS_NV = 8;                 // producer (vector) waves
constexpr int WS_NM = 8;                 // consumer (matrix) waves
constexpr int WS_SLOT_U4 = 15 * 64;      // uint4 per slot: 5 slot groups x 3 pieces x 64 lanes
constexpr int WS_BIAS_FLOATS = 4 * 64;   // four layers x 64 features, activation layout (mlp_common.hpp: idx = (t*2 + h)*16 + r)

struct WsFlags {
    uint32_t ready[WS_NV];       // number of B-fragment sets the V wave has published (sequence number)
    uint32_t bread[WS_NV][2];    // M wave (.., mt): sequence number whose B fragments it has finished reading
    uint32_t done[WS_NV][2];     // M wave (.., mt): sequence number whose accumulator tile is in the slot
    uint32_t abort;              // watchdog: a wait that ran out of patience (a protocol bug, never seen in a correct build)
};

using lds_u32 = __attribute__((address_space(3))) uint32_t;

__device__ __forceinline__ void flag_set(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t flag_get(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#ifndef NSA_WS_SLEEP
#define NSA_WS_SLEEP 1
#endif
// Wait until *p >= v.  Watchdog: a hand-off that does not arrive within ~0.1 s (2^20 polls of >= 64 cycles) can only be a protocol
// bug; instead of hanging the GPU the wave raises `abort`, every other wait of the workgroup then falls through as well, and the
// launch ends with wrong numbers (the bit-identity tests catch that).
constexpr uint32_t WS_PATIENCE = 1u << 20;
__device__ __forceinline__ void flag_wait(const uint32_t* p, uint32_t v, uint32_t* abort) {
    uint32_t spins = 0;
    while (__builtin_amdgcn_readfirstlane(flag_get(p)) < v) {
        __builtin_amdgcn_s_sleep(NSA_WS_SLEEP);
        if ((++spins & 1023u) == 0 && (spins >= WS_PATIENCE || __builtin_amdgcn_readfirstlane(flag_get(abort)))) {
            flag_set(abort, 1u);
            break;
        }
    }
}

#ifdef NSA_X_TS      // profiling build only (tools/ts_profile_ws.py): cycles per phase, per wave, written out at the end of the wave
static __device__ unsigned long long* g_ts_ws = nullptr;
struct WsTs {
    unsigned long long acc[16], prev, start;
    __device__ __forceinline__ void begin() { for (int i = 0; i < 16; ++i) acc[i] = 0; start = prev = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void mark(int slot) { const unsigned long long t = __builtin_readcyclecounter(); acc[slot] += t - prev; prev = t; }
    __device__ __forceinline__ void end() {
        acc[15] = __builtin_readcyclecounter() - start;
        if (g_ts_ws && (threadIdx.x & 63) == 0) {
            unsigned long long* o = g_ts_ws + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16;
            for (int i = 0; i < 16; ++i) o[i] = acc[i];
        }
    }
};
#define WTS_DECL WsTs wts; wts.begin();
#define WTS(slot) wts.mark(slot);
#define WTS_COUNT(slot) wts.acc[slot] += 1;
#define WTS_END wts.end();
#define WTS_ARG , WsTs& wts
#define WTS_PASS , wts
#else
#define WTS_DECL
#define WTS(slot)
#define WTS_COUNT(slot)
#define WTS_END
#define WTS_ARG
#define WTS_PASS
#endif

// tiles of 32 points: V wave v of workgroup b takes tiles (it * G + b) * NV + v
__device__ __forceinline__ uint32_t ws_iters(uint32_t tiles, uint32_t b, uint32_t G, uint32_t v) {
    const uint32_t first = b * WS_NV + v, step = G * WS_NV;
    return first < tiles ? (tiles - first + step - 1) / step : 0u;
}

// ---- M role ---------------------------------------------------------------------------------------------------------------
template <int KS8>
__device__ __forceinline__ void ws_matrix_wave(const float* __restrict__ wblock, int mt, int j, uint32_t tiles, uint4* slots,
                                               const float* bias_lds, WsFlags* fl) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    uint4 a[KS8][3];
    {
        const uint4* __restrict__ w4 = reinterpret_cast<const uint4*>(wblock) + lane;
#pragma unroll
        for (int g = 0; g < KS8; ++g)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) a[g][pc] = w4[((mt * KS8 + g) * 3 + pc) * 64];
    }
    // Clients are served in a FIXED order (tile round `it`, then v): the two waves of a layer must take the same client at the same
    // time -- each waits for the other's reads before it overwrites the slot -- and an order both derive without talking to each
    // other cannot deadlock (a free choice can: wave (j,0) picks client A, wave (j,1) client B, each waits for the other forever).
    // V waves do equal work, so after the first round the fixed order is also the order in which they become ready.
    const lds_u4* bl4 = (const lds_u4*)(bias_lds + j * 64 + (mt * 2 + h) * 16);
    WTS_DECL
    const uint32_t max_it = ws_iters(tiles, blockIdx.x, gridDim.x, 0);
    int n_last = 0;                                   // clients with a tile in the last round (the counts differ by at most one)
#pragma unroll
    for (int v = 0; v < WS_NV; ++v) n_last += ws_iters(tiles, blockIdx.x, gridDim.x, (uint32_t)v) == max_it ? 1 : 0;
    for (uint32_t it = 0; it < max_it; ++it)
#pragma unroll 1
    for (int v = 0; v < (it + 1 < max_it ? WS_NV : n_last); ++v) {
        const uint32_t seq = 4u * it + (uint32_t)j + 1u;
        flag_wait(&fl->ready[v], seq, &fl->abort);
        WTS(0)
        const lds_u4* b4 = (const lds_u4*)(slots + (size_t)v * WS_SLOT_U4) + lane;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 t = bl4[i];
            acc[4 * i + 0] = __uint_as_float(t.x); acc[4 * i + 1] = __uint_as_float(t.y); acc[4 * i + 2] = __uint_as_float(t.z); acc[4 * i + 3] = __uint_as_float(t.w);
        }
        u32x4 nxt[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) nxt[pc] = b4[pc * 64];
#pragma unroll
        for (int g = 0; g < KS8; ++g) {
            u32x4 b[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                b[pc] = nxt[pc];
                if (g + 1 < KS8) nxt[pc] = b4[((g + 1) * 3 + pc) * 64];
            }
            const bf16x8_t bh = as_bf16x8(b[0]), bm = as_bf16x8(b[1]), blo = as_bf16x8(b[2]);
            // the order of mlp_common.hpp::mma_group (smallest terms first): bit-identical accumulation
#define NSA_MM(AP, BV) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[g][AP]), BV, acc, 0, 0, 0);
            NSA_MM(2, bh) NSA_MM(0, blo) NSA_MM(1, bm) NSA_MM(1, bh) NSA_MM(0, bm) NSA_MM(0, bh)
#undef NSA_MM
        }
        // my reads of the slot are complete (the last group's fragments are in registers); the partner's must be too before
        // the accumulators go on top of them
        flag_set(&fl->bread[v][mt], seq);
        WTS(1)
        flag_wait(&fl->bread[v][mt ^ 1], seq, &fl->abort);
        WTS(2)
        lds_u4* o4 = (lds_u4*)(slots + (size_t)v * WS_SLOT_U4) + mt * 4 * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x4 t;
            t.x = __float_as_uint(acc[4 * q + 0]); t.y = __float_as_uint(acc[4 * q + 1]);
            t.z = __float_as_uint(acc[4 * q + 2]); t.w = __float_as_uint(acc[4 * q + 3]);
            o4[q * 64] = t;
        }
        flag_set(&fl->done[v][mt], seq);
        WTS(3)
        WTS_COUNT(14)
    }
    WTS_END
}

// ---- V role ---------------------------------------------------------------------------------------------------------------
template <int KS>
__device__ __forceinline__ void ws_publish(const float (&b)[KS], uint4* slot, int lane, uint32_t* ready, uint32_t seq) {
    constexpr int KS8 = (KS + 7) / 8;
    lds_u4* s4 = (lds_u4*)slot + lane;
#pragma unroll
    for (int g = 0; g < KS8; ++g) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (8 * g + e < KS) ? b[8 * g + e] : 0.0f;
        BFrag f;
        split8(x, f);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            u32x4 t;
            t.x = f.p[pc].x; t.y = f.p[pc].y; t.z = f.p[pc].z; t.w = f.p[pc].w;
            s4[(g * 3 + pc) * 64] = t;
        }
    }
    flag_set(ready, seq);
}

__device__ __forceinline__ void ws_collect(const uint4* slot, int lane, const WsFlags* fl, int v, uint32_t seq, f32x16 (&acc)[2] WTS_ARG) {
    flag_wait(&fl->done[v][0], seq, const_cast<uint32_t*>(&fl->abort));
    flag_wait(&fl->done[v][1], seq, const_cast<uint32_t*>(&fl->abort));
    WTS(3)
    const lds_u4* s4 = (const lds_u4*)slot + lane;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32x4 t = s4[(mt * 4 + q) * 64];
            acc[mt][4 * q + 0] = __uint_as_float(t.x); acc[mt][4 * q + 1] = __uint_as_float(t.y);
            acc[mt][4 * q + 2] = __uint_as_float(t.z); acc[mt][4 * q + 3] = __uint_as_float(t.w);
        }
}

template <int LC, int CC, int LF, int CF>
__device__ __forceinline__ void ws_vector_wave(const SamplerArgs& a, const GridGeom16& gc, const GridGeom16& gf, int v,
                                               uint32_t tiles, uint4* slots, WsFlags* fl) {
    using PC = SdfPack<1>;
    using PF = SdfPack<3>;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const uint64_t total = (uint64_t)a.R * a.E;
    uint4* slot = slots + (size_t)v * WS_SLOT_U4;
    const uint32_t iters = ws_iters(tiles, blockIdx.x, gridDim.x, (uint32_t)v);
    WTS_DECL
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t tile = (it * gridDim.x + blockIdx.x) * WS_NV + (uint32_t)v;
        const uint32_t s0 = 4u * it;
        uint64_t pid = (uint64_t)tile * 32 + (lane & 31);
        const bool live = pid < total;
        if (!live) pid = total - 1;
        const uint32_t ray = ray_of_point(pid, a.E, total);
        const uint32_t idx = (uint32_t)(pid - (uint64_t)ray * a.E);
        RayOfTile rt;
        ray_of_tile(a, ray, rt);
        float x[3], zi, farv;
        sampler_point(a, pid, rt, idx, x, zi, farv);
        float in[SDF_IN_STEPS];
        pe_slots(x, h, in);
        grid_slots<LC, CC, true>(x, a.df_c, a.table_c, gc, h, in);
        WTS(0)
        ws_publish<SDF_IN_STEPS>(in, slot, lane, &fl->ready[v], s0 + 1);           // -> coarse layer 0
        WTS(1)
        grid_slots<LF, CF, true>(x, a.df_f, a.table_f, gf, h, in);                // (while the matrix waves work)
        WTS(2)
        f32x16 acc[2];
        ws_collect(slot, lane, fl, v, s0 + 1, acc WTS_PASS);
        float sdf_c;
        {
            f32x16 ws[2];
            load_vec<2>(a.wp_c + PC::kWSDF, h, ws);
            float part = 0.0f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) part = fmaf(softplus100(acc[mt][r]), ws[mt][r], part);
            sdf_c = xhalf_sum(part) + a.wp_c[PC::kBSDF];
        }
        WTS(4)
        ws_publish<SDF_IN_STEPS>(in, slot, lane, &fl->ready[v], s0 + 2);           // -> fine layer 0
        WTS(5)
        float act[HS];
#pragma unroll
        for (int k = 1; k < 3; ++k) {
            ws_collect(slot, lane, fl, v, s0 + 1 + k, acc WTS_PASS);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) act[16 * mt + r] = softplus100(acc[mt][r]);
            WTS(6)
            ws_publish<HS>(act, slot, lane, &fl->ready[v], s0 + 2 + k);            // -> fine layer k
            WTS(7)
        }
        ws_collect(slot, lane, fl, v, s0 + 4, acc WTS_PASS);
        float sdf_f;
        {
            f32x16 ws[2];
            load_vec<2>(a.wp_f + PF::kWSDF, h, ws);
            float part = 0.0f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) part = fmaf(softplus100(acc[mt][r]), ws[mt][r], part);
            sdf_f = xhalf_sum(part) + a.wp_f[PF::kBSDF];
        }
        if (live && h == 0) {
            a.z[pid] = zi;
            a.sdf[pid] = sdf_c + sdf_f;
            if (idx == 0) a.far[ray] = farv;
        }
        WTS(8)
        WTS_COUNT(14)
    }
    WTS_END
}

#ifndef NSA_WS_MPRIO
#define NSA_WS_MPRIO 2      // static priority of the matrix waves: they issue little and must never queue behind vector code
#endif

template <int LC, int CC, int LF, int CF>
__global__ __launch_bounds__(64 * (WS_NV + WS_NM), 1) void k_sampler_ws(SamplerArgs a, GridGeom16 gc, GridGeom16 gf, uint32_t tiles) {
    extern __shared__ __attribute__((aligned(16))) uint4 ws_smem[];
    uint4* slots = ws_smem;
    float* bias_lds = reinterpret_cast<float*>(ws_smem + WS_NV * WS_SLOT_U4);
    WsFlags* fl = reinterpret_cast<WsFlags*>(bias_lds + WS_BIAS_FLOATS);
    using PC = SdfPack<1>;
    using PF = SdfPack<3>;
    const int wave = threadIdx.x >> 6;
    {   // start-up (the only barrier): flags to zero, the four layers' biases into LDS in activation layout
        uint32_t* f32 = reinterpret_cast<uint32_t*>(fl);
        for (uint32_t i = threadIdx.x; i < sizeof(WsFlags) / 4; i += blockDim.x) f32[i] = 0u;
        if (threadIdx.x < WS_BIAS_FLOATS) {
            const int j = threadIdx.x >> 6, i = threadIdx.x & 63;
            const float* src = j == 0 ? a.wp_c + PC::kB0 : j == 1 ? a.wp_f + PF::kB0 : a.wp_f + PF::bh(j - 1);
            bias_lds[threadIdx.x] = src[i];
        }
        __syncthreads();
    }
    if (wave < WS_NM) {
        // waves w and w + 4 share a SIMD: (layer w, tile 0) sits beside (layer w + 2, tile 1) -- the two halves of a layer run on
        // different SIMDs at the same time, and every SIMD carries one first-layer (30 MFMAs) and one hidden (24) role
        const int mt = wave >> 2;
        const int j = mt ? ((wave & 3) + 2) & 3 : wave;
        if (NSA_WS_MPRIO) __builtin_amdgcn_s_setprio(NSA_WS_MPRIO);
        if (j == 0)      ws_matrix_wave<5>(a.wp_c + PC::kW0, mt, j, tiles, slots, bias_lds, fl);
        else if (j == 1) ws_matrix_wave<5>(a.wp_f + PF::kW0, mt, j, tiles, slots, bias_lds, fl);
        else             ws_matrix_wave<4>(a.wp_f + PF::wh(j - 1), mt, j, tiles, slots, bias_lds, fl);
    } else {
        ws_vector_wave<LC, CC, LF, CF>(a, gc, gf, wave - WS_NM, tiles, slots, fl);
    }
}

}  // namespace nsa

extern "C" {

// Internal (reached through nsa_sampler_sdf when nsa_grid_t.tile == 96; fp32-faithful GEMMs only).
int nsa_sampler_ws_sdf(const float* rays_o, const float* rays_d, uint32_t R, uint32_t E, const float* t_lin, const float* t_rand,
                       float near, float bound, float far_cap, const nsa_grid_t* coarse, const nsa_grid_t* fine,
                       const float* packed_coarse, const float* packed_fine, float* z, float* sdf, float* far,
                       nsa_stream_t stream) {
    using namespace nsa;
    GridGeom16 gc, gf;
    if (int rc = make_grid_geom16(coarse->offsets_host, coarse->L, coarse->S, coarse->H, &gc, coarse->C)) return rc;
    if (int rc = make_grid_geom16(fine->offsets_host, fine->L, fine->S, fine->H, &gf, fine->C)) return rc;
    SamplerArgs a{rays_o, rays_d, t_lin, t_rand, z, sdf, far, R, E, near, bound, far_cap,
                  coarse->table, fine->table, packed_coarse, packed_fine, coarse->divide_factor, fine->divide_factor};
    const uint64_t total = (uint64_t)R * E;
    const uint64_t tiles64 = (total + 31) / 32;
    if (tiles64 > 0x7FFFFFFFull) return NSA_EBADARG;
    const uint32_t tiles = (uint32_t)tiles64;
    static int n_cu = 0;                                 // one persistent workgroup per CU (121 KiB of LDS each)
    if (n_cu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            return NSA_ELAUNCH;
        n_cu = n;
    }
    uint32_t blocks = (tiles + WS_NV - 1) / WS_NV;
    if (blocks > (uint32_t)n_cu) blocks = (uint32_t)n_cu;
    const size_t lds = (size_t)WS_NV * WS_SLOT_U4 * 16 + WS_BIAS_FLOATS * 4 + sizeof(WsFlags);
    auto kern = k_sampler_ws<4, 8, 8, 4>;
    launch_begin();
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return NSA_ELAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * (WS_NV + WS_NM)), lds, (hipStream_t)stream, a, gc, gf, tiles);
    return launch_end();
}

#ifdef NSA_X_TS
int nsa_debug_set_ts_ws(unsigned long long* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(nsa::g_ts_ws), &p, sizeof(p)) == hipSuccess ? 0 : 3;
}
#endif

}  // extern "C"
