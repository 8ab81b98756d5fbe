// render_sampler_sys.hip -- the sampler's SDF pass (reference code/model/ray_sampler.py:90-112,
// code/model/base_networks.py:155-228) as a SYSTOLIC pipeline of specialised waves inside one persistent workgroup per CU.
//
// Second form of the wave-specialised sampler.  The first (render_sampler_ws.hip) kept the whole per-point program in the vector
// waves and lent them matrix waves for the GEMMs; it measured slower (222 vs 178 us): a vector wave held one 32-point tile at a
// time, spent a quarter of it waiting for accumulators to come back, and the matrix waves idled 80 % on flags
// (profiles/r04_ws_sampler_cycle_profile.txt).  Here nothing ever comes BACK to a vector wave:
//
//   V waves (8)          front end only: stratified z, point, positional encoding, both corner gathers; the exact 3-way bf16 split
//                        of the two first-layer inputs; publish  B(coarse layer 0), B(fine layer 0)  -> next tile.  Fire and forget.
//   engine waves (8)     one (layer, 32-feature output tile) each, weight-stationary (the wave's 32 x K block of split weights stays
//                        in 48-60 registers for the whole launch).  A layer engine reads its B fragments, issues the six-product MFMA
//                        groups, applies Softplus to ITS 32 features and
//                          C0 (coarse layer 0)   dots them with the sdf row                      -> sdf_c of the tile
//                          F0, F1 (fine 0, 1)    splits them and writes k-groups {2 mt, 2 mt + 1} of the NEXT layer's B fragments
//                          F2 (fine 2)           dots with the sdf row, adds sdf_c, stores sdf.
//                        The hidden chain never returns to the vector side; the engines' VALU share (softplus + split of 16 values
//                        per lane and tile) runs beside the other engines' MFMAs on the same SIMD.
//
// Every stage handles the tiles in ONE order -- the ticket q a V wave draws when its front end is complete -- so all hand-offs are
// monotone counters in LDS (workgroup-scope release / acquire, no barrier after start-up, a watchdog in every wait) and no stage
// ever waits for a higher ticket: deadlock-free by induction on q.  Buffers (LDS, 142 KiB): 3 + 3 first-layer slots of 15 KiB,
// two rings of 2 x 12 KiB between the fine layers, 16 per-tile records.
// The accumulation order of every fp32 sum is that of k_sampler_sdf (mma_group's product order per accumulator; the sdf-row dot is
// one fma chain over tile 0 then tile 1, handed from the mt = 0 engine to the mt = 1 engine): results are BIT-IDENTICAL.
#include "sampler_common.hpp"

#ifndef NSA_SY_SLEEP
#define NSA_SY_SLEEP 1      // s_sleep argument (x 64 cycles) of a polling wait
#endif
#ifndef NSA_SY_PRIO
#define NSA_SY_PRIO 2       // static priority of the engine waves
#endif

namespace nsa {

constexpr int SY_NV = 8;                 // vector (front-end) waves
constexpr int SY_NE = 8;                 // engine waves: 4 layers x 2 output tiles
#ifndef NSA_SY_N1
#define NSA_SY_N1 3
#endif
#ifndef NSA_SY_NR
#define NSA_SY_NR 2
#endif
constexpr int SY_N1 = NSA_SY_N1;         // first-layer slots per network
constexpr int SY_NR = NSA_SY_NR;         // ring entries between two fine layers
constexpr int SY_NP = 16;                // per-tile records in flight
constexpr int SY_SLOT1_U4 = 15 * 64;     // 5 slot groups x 3 pieces x 64 lanes
constexpr int SY_RING_U4 = 12 * 64;      // 4 slot groups
constexpr int SY_BIAS_FLOATS = 4 * 64;

struct SyFlags {
    uint32_t ticket;                     // next tile sequence number q
    uint32_t abort;                      // watchdog
    uint32_t readyC[SY_N1], readyF[SY_N1];           // = q + 1 once the B fragments of tile q are in the slot
    uint32_t readsC[SY_N1][2], readsF[SY_N1][2];     // engine (.., mt): = q / N1 + 1 once it has read tile q out of the slot
    uint32_t wrote3[SY_NR][2], reads3[SY_NR][2];     // ring F0 -> F1: writer / reader (.., mt): = q / NR + 1
    uint32_t wrote4[SY_NR][2], reads4[SY_NR][2];     // ring F1 -> F2
    uint32_t part_c, part_f;                         // = q + 1: the mt = 0 engine's half of the sdf-row dot of tile q is in part*[]
    uint32_t c0_done, f2_done;                       // = q + 1: sdf_c of tile q is in sdfc[]; tile q is stored
    uint32_t tile_of[SY_NP];                         // 32-point tile index of ticket q (record q % NP)
};

__device__ __forceinline__ void sy_set(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t sy_get(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// wait until *p >= v; ~0.1 s of patience, then the whole workgroup falls through its waits (wrong numbers, never a hang)
__device__ __forceinline__ void sy_wait(const uint32_t* p, uint32_t v, uint32_t* abort) {
    uint32_t spins = 0;
    while (__builtin_amdgcn_readfirstlane(sy_get(p)) < v) {
        __builtin_amdgcn_s_sleep(NSA_SY_SLEEP);
        if ((++spins & 1023u) == 0 && (spins >= (1u << 20) || __builtin_amdgcn_readfirstlane(sy_get(abort)))) {
            sy_set(abort, 1u);
            break;
        }
    }
}

__device__ __forceinline__ uint32_t sy_iters(uint32_t tiles, uint32_t b, uint32_t G, uint32_t v) {
    const uint32_t first = b * SY_NV + v, step = G * SY_NV;
    return first < tiles ? (tiles - first + step - 1) / step : 0u;
}

#ifdef NSA_X_TS      // profiling build only (tools/ts_profile_ws.py --sys): cycles per phase, per wave
static __device__ unsigned long long* g_ts_sys = nullptr;
struct SyTs {
    unsigned long long acc[16], prev, start;
    __device__ __forceinline__ void begin() { for (int i = 0; i < 16; ++i) acc[i] = 0; start = prev = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void mark(int slot) { const unsigned long long t = __builtin_readcyclecounter(); acc[slot] += t - prev; prev = t; }
    __device__ __forceinline__ void end() {
        acc[15] = __builtin_readcyclecounter() - start;
        if (g_ts_sys && (threadIdx.x & 63) == 0) {
            unsigned long long* o = g_ts_sys + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16;
            for (int i = 0; i < 16; ++i) o[i] = acc[i];
        }
    }
};
// per-tile event trace of workgroup 0: trace[q * 32 + event] = cycle counter
static __device__ unsigned long long* g_tr_sys = nullptr;
__device__ __forceinline__ void ytrace(uint32_t q, int ev) {
    if (g_tr_sys && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && q < 96) g_tr_sys[q * 32 + ev] = __builtin_readcyclecounter();
}
#define YTR(q, ev) ytrace(q, ev);
#define YTS_DECL SyTs yts; yts.begin();
#define YTS(slot) yts.mark(slot);
#define YTS_COUNT(slot) yts.acc[slot] += 1;
#define YTS_END yts.end();
#else
#define YTR(q, ev)
#define YTS_DECL
#define YTS(slot)
#define YTS_COUNT(slot)
#define YTS_END
#endif

struct SyLds {
    uint4* slotC;      // [N1][SLOT1]
    uint4* slotF;      // [N1][SLOT1]
    uint4* ring3;      // [NR][RING]
    uint4* ring4;      // [NR][RING]
    float* bias;       // [4][64] activation layout
    float* part_c;     // [NP][64]   mt = 0 half of the coarse sdf-row dot
    float* part_f;     // [NP][64]
    float* sdfc;       // [NP][64]   coarse sdf of the tile's points (lane p and lane p + 32 hold the same value)
    SyFlags* fl;
};

// eight fp32 values of this lane -> three 16-byte bf16 fragments of one slot group, written to dst[(piece) * 64] (dst includes + lane)
__device__ __forceinline__ void sy_put_group(const float (&x)[8], lds_u4* dst) {
    BFrag f;
    split8(x, f);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
        u32x4 t;
        t.x = f.p[pc].x; t.y = f.p[pc].y; t.z = f.p[pc].z; t.w = f.p[pc].w;
        dst[pc * 64] = t;
    }
}

// ---- engines --------------------------------------------------------------------------------------------------------------------
// STAGE 0: coarse layer 0   1: fine layer 0   2: fine layer 1   3: fine layer 2
template <int STAGE>
__device__ __forceinline__ void sy_engine(const SamplerArgs& a, int mt, uint32_t n_tiles_wg, uint32_t tiles, const SyLds& L) {
    using PC = SdfPack<1>;
    using PF = SdfPack<3>;
    constexpr int KS8 = STAGE < 2 ? 5 : 4;
    constexpr bool DOT = STAGE == 0 || STAGE == 3;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    SyFlags* fl = L.fl;
    const float* wblock = STAGE == 0 ? a.wp_c + PC::kW0 : STAGE == 1 ? a.wp_f + PF::kW0 : a.wp_f + PF::wh(STAGE - 1);
    uint4 aw[KS8][3];
    {
        const uint4* __restrict__ w4 = reinterpret_cast<const uint4*>(wblock) + lane;
#pragma unroll
        for (int g = 0; g < KS8; ++g)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) aw[g][pc] = w4[((mt * KS8 + g) * 3 + pc) * 64];
    }
    float ws[16];                                    // this tile's 32 rows of the sdf row, activation layout (engines C0 / F2 only)
    float bias_out = 0.0f;
    if (DOT) {
        const float* wv = (STAGE == 0 ? a.wp_c + PC::kWSDF : a.wp_f + PF::kWSDF) + (mt * 2 + h) * 16;
#pragma unroll
        for (int r = 0; r < 16; ++r) ws[r] = wv[r];
        bias_out = STAGE == 0 ? a.wp_c[PC::kBSDF] : a.wp_f[PF::kBSDF];
    }
    const lds_u4* bl4 = (const lds_u4*)(L.bias + STAGE * 64 + (mt * 2 + h) * 16);
    const uint64_t total = (uint64_t)a.R * a.E;
    YTS_DECL
    for (uint32_t q = 0; q < n_tiles_wg; ++q) {
        // ---- B fragments of tile q for this layer
        const lds_u4* b4;
        if (STAGE == 0)      { sy_wait(&fl->readyC[q % SY_N1], q + 1, &fl->abort); b4 = (const lds_u4*)(L.slotC + (q % SY_N1) * SY_SLOT1_U4) + lane; }
        else if (STAGE == 1) { sy_wait(&fl->readyF[q % SY_N1], q + 1, &fl->abort); b4 = (const lds_u4*)(L.slotF + (q % SY_N1) * SY_SLOT1_U4) + lane; }
        else if (STAGE == 2) {
            sy_wait(&fl->wrote3[q % SY_NR][0], q / SY_NR + 1, &fl->abort);
            sy_wait(&fl->wrote3[q % SY_NR][1], q / SY_NR + 1, &fl->abort);
            b4 = (const lds_u4*)(L.ring3 + (q % SY_NR) * SY_RING_U4) + lane;
        } else {
            sy_wait(&fl->wrote4[q % SY_NR][0], q / SY_NR + 1, &fl->abort);
            sy_wait(&fl->wrote4[q % SY_NR][1], q / SY_NR + 1, &fl->abort);
            b4 = (const lds_u4*)(L.ring4 + (q % SY_NR) * SY_RING_U4) + lane;
        }
        YTS(0)
        YTR(q, 8 + STAGE * 4 + mt * 2)
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 t = bl4[i];
            acc[4 * i + 0] = __uint_as_float(t.x); acc[4 * i + 1] = __uint_as_float(t.y);
            acc[4 * i + 2] = __uint_as_float(t.z); acc[4 * i + 3] = __uint_as_float(t.w);
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
#define NSA_MM(AP, BV) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(aw[g][AP]), BV, acc, 0, 0, 0);
            NSA_MM(2, bh) NSA_MM(0, blo) NSA_MM(1, bm) NSA_MM(1, bh) NSA_MM(0, bm) NSA_MM(0, bh)       // mma_group's order
#undef NSA_MM
        }
        // the fragments are in registers: hand the buffer back
        if (STAGE == 0)      sy_set(&fl->readsC[q % SY_N1][mt], q / SY_N1 + 1);
        else if (STAGE == 1) sy_set(&fl->readsF[q % SY_N1][mt], q / SY_N1 + 1);
        else if (STAGE == 2) sy_set(&fl->reads3[q % SY_NR][mt], q / SY_NR + 1);
        else                 sy_set(&fl->reads4[q % SY_NR][mt], q / SY_NR + 1);
        YTS(1)
        // ---- this engine's 32 features
        if (DOT) {
            float* part_buf = (STAGE == 0 ? L.part_c : L.part_f) + (q % SY_NP) * 64 + lane;
            uint32_t* part_flag = STAGE == 0 ? &fl->part_c : &fl->part_f;
            float sp[16];                            // (the transcendentals before the wait: only the 16 fmas depend on the partner)
#pragma unroll
            for (int r = 0; r < 16; ++r) sp[r] = softplus100(acc[r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(sp[r]));
            YTS(3)
            float part = 0.0f;
            if (mt == 1) {                           // continue the mt = 0 engine's fma chain (one chain over both tiles, as sdf_only)
                sy_wait(part_flag, q + 1, &fl->abort);
                part = *(const __attribute__((address_space(3))) float*)part_buf;
            }
            YTS(2)
#pragma unroll
            for (int r = 0; r < 16; ++r) part = fmaf(sp[r], ws[r], part);
            if (mt == 0) {
                // (record q % NP was last used by tile q - NP, whose chain the mt = 1 engine finished long ago: it is NP tiles behind)
                *(__attribute__((address_space(3))) float*)part_buf = part;
                sy_set(part_flag, q + 1);
            } else {
                const float s = xhalf_sum(part) + bias_out;
                if (STAGE == 0) {
                    *(__attribute__((address_space(3))) float*)(L.sdfc + (q % SY_NP) * 64 + lane) = s;
                    sy_set(&fl->c0_done, q + 1);
                } else {
                    YTS(3)
                    sy_wait(&fl->c0_done, q + 1, &fl->abort);
                    YTS(2)
                    const float sc = *(const __attribute__((address_space(3))) float*)(L.sdfc + (q % SY_NP) * 64 + lane);
                    const uint32_t tile = __builtin_amdgcn_readfirstlane(sy_get(&fl->tile_of[q % SY_NP]));
                    const uint64_t pid = (uint64_t)tile * 32 + (lane & 31);
                    if (pid < total && h == 0) a.sdf[pid] = sc + s;
                    sy_set(&fl->f2_done, q + 1);
                }
            }
        } else {
            float act[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) act[r] = softplus100(acc[r]);
            uint4* ring = STAGE == 1 ? L.ring3 : L.ring4;
            uint32_t (*reads)[2] = STAGE == 1 ? fl->reads3 : fl->reads4;
            uint32_t (*wrote)[2] = STAGE == 1 ? fl->wrote3 : fl->wrote4;
            const uint32_t e = q % SY_NR;
            YTS(3)
            if (q >= SY_NR) {                        // the previous occupant (tile q - NR) has been read by both engines of the next layer
                sy_wait(&reads[e][0], q / SY_NR, &fl->abort);
                sy_wait(&reads[e][1], q / SY_NR, &fl->abort);
            }
            YTS(2)
            lds_u4* dst = (lds_u4*)(ring + e * SY_RING_U4) + lane;
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {         // hidden slot s = 16 mt + r  ->  k-group 2 mt + (r >> 3)
                float x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] = act[8 * gg + k];
                sy_put_group(x, dst + ((2 * mt + gg) * 3) * 64);
            }
            sy_set(&wrote[e][mt], q / SY_NR + 1);
        }
        YTS(3)
        YTR(q, 8 + STAGE * 4 + mt * 2 + 1)
        YTS_COUNT(14)
    }
    YTS_END
}

// ---- front end -----------------------------------------------------------------------------------------------------------------
template <int LC, int CC, int LF, int CF>
__device__ __forceinline__ void sy_vector_wave(const SamplerArgs& a, const GridGeom16& gc, const GridGeom16& gf, int v,
                                               uint32_t tiles, const SyLds& L) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    SyFlags* fl = L.fl;
    const uint64_t total = (uint64_t)a.R * a.E;
    const uint32_t iters = sy_iters(tiles, blockIdx.x, gridDim.x, (uint32_t)v);
    YTS_DECL
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t tile = (it * gridDim.x + blockIdx.x) * SY_NV + (uint32_t)v;
        uint64_t pid = (uint64_t)tile * 32 + (lane & 31);
        const bool live = pid < total;
        if (!live) pid = total - 1;
        const uint32_t ray = ray_of_point(pid, a.E, total);
        const uint32_t idx = (uint32_t)(pid - (uint64_t)ray * a.E);
        RayOfTile rt;
        ray_of_tile(a, ray, rt);
        float x[3], zi, farv;
        sampler_point(a, pid, rt, idx, x, zi, farv);
        // order chosen for register pressure (128 per lane at 16 waves per CU): the fine gather first (32 corner registers at a time,
        // leaves 16 values), then the coarse gather (64 corner registers), the positional encoding (no memory) last
        float in[SDF_IN_STEPS], fine[16];
        YTS(0)
        {
            float tmp[SDF_IN_STEPS];
            grid_slots<LF, CF, true>(x, a.df_f, a.table_f, gf, h, tmp);
#pragma unroll
            for (int s = 0; s < 16; ++s) fine[s] = tmp[20 + s];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) asm volatile("" ::"v"(fine[s]));
        YTS(3)
        grid_slots<LC, CC, true>(x, a.df_c, a.table_c, gc, h, in);
#pragma unroll
        for (int s = 20; s < SDF_IN_STEPS; ++s) asm volatile("" ::"v"(in[s]));
        YTS(2)
        {
            float pe[SDF_IN_STEPS];
            pe_slots(x, h, pe);
#pragma unroll
            for (int s = 0; s < 20; ++s) in[s] = pe[s];
        }
        YTS(1)
        if (live && h == 0) {
            a.z[pid] = zi;
            if (idx == 0) a.far[ray] = farv;
        }
        // Everything the two publishes need must be IN REGISTERS before the ticket is drawn: without this fence the compiler sinks the
        // fine gather's blend (and its wait for memory) below the ticket, and a tile that holds its place in every stage's order then
        // sits on a load for ~12 k cycles (measured, per-tile trace of tools/ts_profile_ws.py --sys --trace: 225 -> see DESIGN 4.1).
#pragma unroll
        for (int s = 0; s < SDF_IN_STEPS; ++s) asm volatile("" ::"v"(in[s]));
#pragma unroll
        for (int s = 0; s < 16; ++s) asm volatile("" ::"v"(fine[s]));
        __builtin_amdgcn_sched_barrier(0);
        YTS(3)
        // ---- ticket: from here on this tile is number q in every stage's order
        uint32_t q = 0;
        if (lane == 0) q = __hip_atomic_fetch_add(&fl->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        q = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
        YTR(q, 0)
        const uint32_t s1 = q % SY_N1;
        if (q >= SY_NP) sy_wait(&fl->f2_done, q - SY_NP + 1, &fl->abort);      // record q % NP is free again
        YTS(6)
        if (lane == 0) fl->tile_of[q % SY_NP] = tile;
        if (q >= SY_N1) {                                                       // slot of tile q - N1: read by both coarse engines
            sy_wait(&fl->readsC[s1][0], q / SY_N1, &fl->abort);
            sy_wait(&fl->readsC[s1][1], q / SY_N1, &fl->abort);
        }
        YTS(4)
        YTR(q, 1)
        // slot groups 0 and 1 (slots 0..15: position and positional encoding) are the same for both networks: split once
        BFrag pe_frag[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float xx[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) xx[e] = in[8 * g + e];
            split8(xx, pe_frag[g]);
        }
        auto put_frag = [](const BFrag& f, lds_u4* dst) {
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                u32x4 t;
                t.x = f.p[pc].x; t.y = f.p[pc].y; t.z = f.p[pc].z; t.w = f.p[pc].w;
                dst[pc * 64] = t;
            }
        };
        {
            lds_u4* dst = (lds_u4*)(L.slotC + s1 * SY_SLOT1_U4) + lane;
            put_frag(pe_frag[0], dst);
            put_frag(pe_frag[1], dst + 3 * 64);
#pragma unroll
            for (int g = 2; g < 5; ++g) {
                float xx[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) xx[e] = (8 * g + e < SDF_IN_STEPS) ? in[8 * g + e] : 0.0f;
                sy_put_group(xx, dst + (g * 3) * 64);
            }
            sy_set(&fl->readyC[s1], q + 1);
        }
        YTR(q, 2)
        YTS(5)
#pragma unroll
        for (int s = 0; s < 16; ++s) in[20 + s] = fine[s];
        if (q >= SY_N1) {
            sy_wait(&fl->readsF[s1][0], q / SY_N1, &fl->abort);
            sy_wait(&fl->readsF[s1][1], q / SY_N1, &fl->abort);
        }
        YTS(7)
        YTR(q, 3)
        {
            lds_u4* dst = (lds_u4*)(L.slotF + s1 * SY_SLOT1_U4) + lane;
            put_frag(pe_frag[0], dst);
            put_frag(pe_frag[1], dst + 3 * 64);
#pragma unroll
            for (int g = 2; g < 5; ++g) {
                float xx[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) xx[e] = (8 * g + e < SDF_IN_STEPS) ? in[8 * g + e] : 0.0f;
                sy_put_group(xx, dst + (g * 3) * 64);
            }
            sy_set(&fl->readyF[s1], q + 1);
        }
        YTR(q, 4)
        YTS(5)
        YTS_COUNT(14)
    }
    YTS_END
}

template <int LC, int CC, int LF, int CF>
__global__ __launch_bounds__(64 * (SY_NV + SY_NE), 1) void k_sampler_sys(SamplerArgs a, GridGeom16 gc, GridGeom16 gf, uint32_t tiles) {
    extern __shared__ __attribute__((aligned(16))) uint4 sy_smem[];
    using PC = SdfPack<1>;
    using PF = SdfPack<3>;
    SyLds L;
    L.slotC = sy_smem;
    L.slotF = L.slotC + SY_N1 * SY_SLOT1_U4;
    L.ring3 = L.slotF + SY_N1 * SY_SLOT1_U4;
    L.ring4 = L.ring3 + SY_NR * SY_RING_U4;
    L.bias = reinterpret_cast<float*>(L.ring4 + SY_NR * SY_RING_U4);
    L.part_c = L.bias + SY_BIAS_FLOATS;
    L.part_f = L.part_c + SY_NP * 64;
    L.sdfc = L.part_f + SY_NP * 64;
    L.fl = reinterpret_cast<SyFlags*>(L.sdfc + SY_NP * 64);
    const int wave = threadIdx.x >> 6;
    {   // start-up (the only barrier): flags to zero, the four layers' biases into LDS in activation layout
        uint32_t* f32 = reinterpret_cast<uint32_t*>(L.fl);
        for (uint32_t i = threadIdx.x; i < sizeof(SyFlags) / 4; i += blockDim.x) f32[i] = 0u;
        if (threadIdx.x < SY_BIAS_FLOATS) {
            const int j = threadIdx.x >> 6, i = threadIdx.x & 63;
            const float* src = j == 0 ? a.wp_c + PC::kB0 : j == 1 ? a.wp_f + PF::kB0 : a.wp_f + PF::bh(j - 1);
            L.bias[threadIdx.x] = src[i];
        }
        __syncthreads();
    }
    uint32_t n_wg = 0;
#pragma unroll
    for (int v = 0; v < SY_NV; ++v) n_wg += sy_iters(tiles, blockIdx.x, gridDim.x, (uint32_t)v);
    if (wave < SY_NE) {
        // waves w and w + 4 share a SIMD: (layer w, tile 0) sits beside (layer w + 2, tile 1)
        const int mt = wave >> 2;
        const int stage = mt ? ((wave & 3) + 2) & 3 : wave;
        if (NSA_SY_PRIO) __builtin_amdgcn_s_setprio(NSA_SY_PRIO);
        if (stage == 0)      sy_engine<0>(a, mt, n_wg, tiles, L);
        else if (stage == 1) sy_engine<1>(a, mt, n_wg, tiles, L);
        else if (stage == 2) sy_engine<2>(a, mt, n_wg, tiles, L);
        else                 sy_engine<3>(a, mt, n_wg, tiles, L);
    } else {
        sy_vector_wave<LC, CC, LF, CF>(a, gc, gf, wave - SY_NE, tiles, L);
    }
}

}  // namespace nsa

extern "C" {

// Internal (reached through nsa_sampler_sdf when nsa_grid_t.tile == 97; fp32-faithful GEMMs only).
int nsa_sampler_sys_sdf(const float* rays_o, const float* rays_d, uint32_t R, uint32_t E, const float* t_lin, const float* t_rand,
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
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            return NSA_ELAUNCH;
        n_cu = n;
    }
    uint32_t blocks = (tiles + SY_NV - 1) / SY_NV;
    if (blocks > (uint32_t)n_cu) blocks = (uint32_t)n_cu;
    const size_t lds = ((size_t)2 * SY_N1 * SY_SLOT1_U4 + (size_t)2 * SY_NR * SY_RING_U4) * 16 + (SY_BIAS_FLOATS + 3 * SY_NP * 64) * 4 +
                       sizeof(SyFlags);
    auto kern = k_sampler_sys<4, 8, 8, 4>;
    launch_begin();
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return NSA_ELAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * (SY_NV + SY_NE)), lds, (hipStream_t)stream, a, gc, gf, tiles);
    return launch_end();
}

#ifdef NSA_X_TS
int nsa_debug_set_ts_sys(unsigned long long* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(nsa::g_ts_sys), &p, sizeof(p)) == hipSuccess ? 0 : 3;
}
int nsa_debug_set_trace_sys(unsigned long long* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(nsa::g_tr_sys), &p, sizeof(p)) == hipSuccess ? 0 : 3;
}
#endif

}  // extern "C"
