// grid_common.hpp -- level geometry + per-(point,level) cell location shared by every gfx950 kernel
// that touches a multi-resolution grid.  Algorithm per reference code/hashencoder/src/hashencoder.cu
// (index rule :35-73, geometry :179-181, cell split :188-195); the host derives everything that is
// point-independent once per call and hands it to the kernels as kernel arguments (SGPR loads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/nicer_slam_amd.h"

namespace nsa {

// flags
constexpr uint32_t LV_HASHED = 1u;    // index = xor-prime hash of the cell
constexpr uint32_t LV_POW2 = 2u;      // rows is a power of two: modulo == mask
constexpr uint32_t LV_SUBONCE = 4u;   // dense and idx < 2*rows guaranteed: modulo == one conditional subtract
constexpr uint32_t LV_GENERIC = 8u;   // neither: needs a true uint32 modulo (never the case for the shipped grids)
// fast corner addressing of the fused kernels (gather_corners, D = 3): byte offsets, three adds for the eight corners
constexpr uint32_t LV_FASTDENSE = 16u;  // dense, no uint32 stride wrap, idx < 2*rows, all byte offsets < 2^31
constexpr uint32_t LV_FASTHASH = 32u;   // hashed with a power-of-two row count, all byte offsets < 2^32

struct LevelGeom {
    float scale;     // exp2f(level*S)*H - 1   (float32, hashencoder.cu:180)
    uint32_t rows;   // hashmap_size = offsets[l+1]-offsets[l]
    uint32_t row0;   // offsets[l]
    uint32_t s1;     // dense stride of dim 1 (= res, uint32)
    uint32_t s2;     // dense stride of dim 2 (= res*res, uint32 wrap)
    uint32_t flags;
    uint32_t mask;   // rows-1 when rows is a power of two, else 0xFFFFFFFF (so `idx & mask` is always legal)
    // byte-scaled copies for the fast corner addressing of the fused kernels (corner_offsets; B = C * 4 bytes per row)
    uint32_t row0B;  // row0 * B
    uint32_t s1B;    // s1 * B
    uint32_t s2B;    // s2 * B
    uint32_t limB;   // LV_FASTDENSE: (row0 + rows) * B - (s2B + s1B + B): corner (0,0,0) below it <=> all eight rows inside the level
    uint32_t pad;
};

struct GridGeom {
    LevelGeom lv[NSA_MAX_LEVELS];
};

// Host: emulate get_grid_index's stride loop (hashencoder.cu:56-70) in uint32 to classify the level.
inline int make_grid_geom(const int32_t* offsets_host, uint32_t L, uint32_t D, float S, uint32_t H, GridGeom* out,
                          uint32_t C = 8) {
    if (L > NSA_MAX_LEVELS) return NSA_ETOO_MANY_LEVELS;
    const uint64_t table_bytes = (uint64_t)(uint32_t)offsets_host[L] * C * 4;
    for (uint32_t l = 0; l < L; ++l) {
        LevelGeom g;
        g.scale = exp2f((float)l * S) * (float)H - 1.0f;
        const uint32_t res = (uint32_t)ceilf(g.scale) + 1u;
        g.row0 = (uint32_t)offsets_host[l];
        g.rows = (uint32_t)(offsets_host[l + 1] - offsets_host[l]);
        if (g.rows == 0) return NSA_EBADARG;
        uint32_t stride = 1u;
        uint64_t max_idx = 0;     // exact (non-wrapping) bound of the dense index, to pick LV_SUBONCE
        bool wrapped = false;
        uint32_t d = 0;
        for (; d < D && stride <= g.rows; ++d) {
            max_idx += (uint64_t)res * stride;     // cell+1 <= res
            uint64_t next = (uint64_t)stride * res;
            if (next > 0xFFFFFFFFull) wrapped = true;
            stride = (uint32_t)next;
        }
        g.s1 = res;
        g.s2 = res * res;
        g.flags = 0;
        if (stride > g.rows) g.flags |= LV_HASHED;
        if ((g.rows & (g.rows - 1)) == 0) g.flags |= LV_POW2;
        else if (!(g.flags & LV_HASHED) && !wrapped && max_idx < 2ull * g.rows) g.flags |= LV_SUBONCE;
        else g.flags |= LV_GENERIC;
        g.mask = (g.flags & LV_POW2) ? g.rows - 1u : 0xFFFFFFFFu;
        g.pad = 0;
        const uint32_t B = C * 4;
        g.row0B = g.row0 * B;
        g.s1B = g.s1 * B;
        g.s2B = g.s2 * B;
        g.limB = 0;
        if (D == 3 && table_bytes < (1ull << 32)) {
            const uint64_t end_bytes = ((uint64_t)g.row0 + g.rows) * B;
            if (!(g.flags & LV_HASHED) && !wrapped && max_idx < 2ull * g.rows && end_bytes < (1ull << 31) &&
                (uint64_t)g.s2 * B < (1ull << 24) && res < (1u << 20)) {
                g.flags |= LV_FASTDENSE;
                g.limB = (uint32_t)end_bytes - (g.s2B + g.s1B + B);
            }
            if ((g.flags & LV_HASHED) && (g.flags & LV_POW2)) g.flags |= LV_FASTHASH;
        }
        out->lv[l] = g;
    }
    return NSA_OK;
}

inline bool has_generic_level(const LevelGeom* lv, uint32_t L) {
    for (uint32_t l = 0; l < L; ++l)
        if (lv[l].flags & LV_GENERIC) return true;
    return false;
}

// hipGetLastError() also reports stale non-errors left by OTHER runtime calls on this host thread (e.g. PyTorch's
// hipEventQuery -> hipErrorNotReady), so every entry point clears it before launching and reads it after.
inline void launch_begin() { (void)hipGetLastError(); }
inline int launch_end() { return hipGetLastError() == hipSuccess ? NSA_OK : NSA_ELAUNCH; }

// a*b rounded on its own (never contracted into an FMA): sample positions are built as o + z*d by two separate torch
// kernels in the reference (network.py:112-114, ray_sampler.py:97); with the `far` sample sitting exactly on the cube
// face by construction, a fused multiply-add can flip that point's in-range test (hashencoder.cu:155-159).
__device__ __forceinline__ float mul_rn(float a, float b) {
    float r;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---- device side -------------------------------------------------------------------------------
// [-1,1] -> [0,1] map of HashEncoder.forward: (x / divide_factor + 1) / 2 (hashgrid.py:203, size = 1).  (A scalar branch that skips
// the IEEE division sequence when divide_factor == 1 -- every shipped configuration -- was measured: no gain in any kernel, +4 us in
// the colour forward, whose first gathers then wait behind the branch; profiles/r03_ab_experiments.txt.)
__device__ __forceinline__ float to_unit(float x, float divide_factor) { return (x / divide_factor + 1.0f) / 2.0f; }

template <int D>
__device__ __forceinline__ uint32_t level_row(const LevelGeom& g, const uint32_t (&q)[D]) {
    uint32_t idx;
    if (g.flags & LV_HASHED) {                        // wave-uniform branch (level is per block)
        idx = q[0];                                   // prime[0] == 1
        if (D > 1) idx ^= q[1] * 2654435761u;
        if (D > 2) idx ^= q[2] * 805459861u;
    } else {
        idx = q[0];
        if (D > 1) idx += q[1] * g.s1;
        if (D > 2) idx += q[2] * g.s2;
    }
    if (g.flags & LV_POW2) return idx & (g.rows - 1u);
    if (g.flags & LV_SUBONCE) return idx >= g.rows ? idx - g.rows : idx;
    return idx % g.rows;
}

// Run-merged, row-coalesced atomic scatter of C-channel rows.
// Measured on MI355X (tools/micro/atomic_bench.hip): float atomics retire at ~20 G REQUESTS/s chip-wide, independent of
// table size and of contention, where one request = the lanes of one instruction that fall into the same row (cache
// line segment).  A lane-per-row issue (lane l adds channel c of ITS row, C instructions) therefore costs C requests
// per row; a lane-per-channel issue (C adjacent lanes add the C channels of ONE row) costs one: 8x / 4x / 2x fewer
// requests for C = 8 / 4 / 2.  Two steps:
//  1. run merge: the points of a wave are consecutive samples of a ray and walk through each cell of a coarse level in
//     RUNS of equal row index; the lanes of a run are summed with a segmented shuffle scan and only the run's last lane
//     keeps a row to send.
//  2. transposed issue through a per-wave LDS tile: in round j, lane group g (C lanes) sends the row held by lane
//     g*C + j, lane % C = channel.
// `key` = destination row (relative to `table`) or 0xFFFFFFFF for lanes with nothing to add (they never merge).
// Blocks using this are 256 threads (4 waves).
//  3. wider spans: a request may cover at least 64 contiguous bytes (same benchmark: C lanes on one 8/16/32-byte row and
//     2C lanes on two ADJACENT rows both retire at ~20 G requests/s, i.e. 41 G rows/s for pairs).  The two x-neighbour
//     corners of a cell are adjacent rows on a dense level, and on a hashed level for every cell with even x: they are sent
//     as one 2C-float span (scatter_x_pair).
// `elem` = float offset of the span's first channel in `table`, or 0xFFFFFFFF for lanes with nothing to add.
// `tile` = this wave's 64*(W+1)-float LDS scratch.  Must be called by all 64 lanes (shuffles / ballots inside).
template <int W>
__device__ __forceinline__ void scatter_span(float* __restrict__ table, uint32_t elem, float (&val)[W], int lane, float* tile) {
    const uint32_t prev = __shfl_up(elem, 1);
    const bool head = lane == 0 || prev != elem || elem == 0xFFFFFFFFu;
    const unsigned long long hm = __ballot(head);
    const int seg = __popcll(hm & ((2ull << lane) - 1ull));            // inclusive count of heads = segment id
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int oseg = __shfl_up(seg, off);
        const bool take = lane >= off && oseg == seg;
#pragma unroll
        for (int c = 0; c < W; ++c) {
            const float o = __shfl_up(val[c], off);
            if (take) val[c] += o;
        }
    }
    const bool tail = lane == 63 || ((hm >> (lane + 1)) & 1ull);
    const uint32_t send = tail ? elem : 0xFFFFFFFFu;
    if (W == 1) {
        if (send != 0xFFFFFFFFu) atomicAdd(table + send, val[0]);      // -munsafe-fp-atomics: global_atomic_add_f32
        return;
    }
#pragma unroll
    for (int c = 0; c < W; ++c) tile[lane * (W + 1) + c] = val[c];     // row pitch W+1: conflict-free transposed reads
    __builtin_amdgcn_wave_barrier();                                   // LDS ops of one wave execute in order
    // round j sends the spans held by the CONTIGUOUS lanes [j * 64/W, (j+1) * 64/W): neighbouring lanes are neighbouring points
    // (consecutive samples of a ray / Morton order), whose spans often share a 64-byte segment and then leave as one request
    const int slot = lane / W, ch = lane % W;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int src = j * (64 / W) + slot;
        const uint32_t e = __shfl(send, src);
        const float v = tile[src * (W + 1) + ch];
        if (e != 0xFFFFFFFFu) atomicAdd(table + (size_t)e + ch, v);
    }
    __builtin_amdgcn_wave_barrier();
}

// one C-channel row per lane; `key` = row index or 0xFFFFFFFF
template <int C>
__device__ __forceinline__ void scatter_runs(float* __restrict__ table, uint32_t key, float (&val)[C], int lane, float* tile) {
    scatter_span<C>(table, key == 0xFFFFFFFFu ? key : key * (uint32_t)C, val, lane, tile);
}

template <int C>
__device__ __forceinline__ void scatter_runs(float* __restrict__ table, uint32_t key, float (&val)[C], int lane) {
    __shared__ float stage[4][64 * (C + 1)];
    scatter_runs<C>(table, key, val, lane, stage[threadIdx.x >> 6]);
}

// The two x-neighbour corner rows (r0: corner x, r1: corner x + 1) of a cell as ONE 2C-float span when they are adjacent in
// memory: always on a dense level (r1 == r0 + 1; except where the level's index wraps), and on a HASHED level whenever the
// cell's x is even -- the hash xors x in with prime 1, so the two rows then differ in bit 0 only (either order).  Lanes whose
// rows are not adjacent send them as two single rows; that path is skipped when no lane of the wave needs it.
// `tile` holds 64*(2C+1) floats.  Must be called by all 64 lanes.
template <int C>
__device__ __forceinline__ void scatter_x_pair(float* __restrict__ table, uint32_t r0, uint32_t r1, bool valid,
                                               const float (&v0)[C], const float (&v1)[C], int lane, float* tile) {
    const bool up = r1 == r0 + 1u, down = r0 == r1 + 1u;
    const bool adj = valid && (up || down);
    float val[2 * C];
#pragma unroll
    for (int c = 0; c < C; ++c) { val[c] = down ? v1[c] : v0[c]; val[C + c] = down ? v0[c] : v1[c]; }
    scatter_span<2 * C>(table, adj ? (down ? r1 : r0) * (uint32_t)C : 0xFFFFFFFFu, val, lane, tile);
    const bool single = valid && !adj;
    if (__ballot(single)) {                                            // wave-uniform
        float a[C], b[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { a[c] = v0[c]; b[c] = v1[c]; }
        scatter_runs<C>(table, single ? r0 : 0xFFFFFFFFu, a, lane, tile);
        scatter_runs<C>(table, single ? r1 : 0xFFFFFFFFu, b, lane, tile);
    }
}

template <int C>
__device__ __forceinline__ void scatter_x_pair(float* __restrict__ table, uint32_t r0, uint32_t r1, bool valid,
                                               const float (&v0)[C], const float (&v1)[C], int lane) {
    __shared__ float stage2[4][64 * (2 * C + 1)];
    scatter_x_pair<C>(table, r0, r1, valid, v0, v1, lane, stage2[threadIdx.x >> 6]);
}

// Range test + cell/fraction split.  Returns false for a point outside [0,1]^D (NaN passes, as in the
// reference's `<`/`>` tests).  w = smoothstep(t), dw = smoothstep'(t).
template <int D>
__device__ __forceinline__ bool locate(const float (&x)[D], float scale, uint32_t (&cell)[D], float (&w)[D], float (&dw)[D]) {
    bool inside = true;
#pragma unroll
    for (int d = 0; d < D; ++d) inside = inside && !(x[d] < 0.0f || x[d] > 1.0f);
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float p = x[d] * scale;
        // cell = (uint32_t)floorf(p), t = p - (float)cell (hashencoder.cu:188-195).  For every in-range point p >= 0, so the
        // truncating conversion IS the floor and p - floor(p) is v_fract_f32 (the subtraction is exact in fp32); results of
        // out-of-range points are discarded by every caller (`inside`).
        cell[d] = (uint32_t)p;
        const float t = __builtin_amdgcn_fractf(p);
        dw[d] = 6.0f * t * (1.0f - t);
        w[d] = t * t * (3.0f - 2.0f * t);
    }
    return inside;
}

template <int C> struct VecOf;
template <> struct VecOf<1> { using type = float; };
template <> struct VecOf<2> { using type = float2; };
template <> struct VecOf<4> { using type = float4; };

// C contiguous floats starting at a C*4-byte aligned address -> registers (1 or 2 vector loads).
template <int C>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&v)[C]) {
    if constexpr (C == 8) {
        const float4 a = reinterpret_cast<const float4*>(p)[0];
        const float4 b = reinterpret_cast<const float4*>(p)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else if constexpr (C == 4) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    } else if constexpr (C == 2) {
        const float2 a = *reinterpret_cast<const float2*>(p);
        v[0] = a.x; v[1] = a.y;
    } else {
        v[0] = p[0];
    }
}

template <int C>
__device__ __forceinline__ void store_row(float* __restrict__ p, const float (&v)[C]) {
    if constexpr (C == 8) {
        reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
    } else if constexpr (C == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (C == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
    } else {
        p[0] = v[0];
    }
}

// Gather the 2^D corner rows of the cell into registers: corner bit d set <=> +1 along dim d.
//
// Generic form (stand-alone operator, and the fall-back of the fast form): branch-free -- the level, hence hashed/dense, may
// differ between the lanes of a wave -- per dimension the two candidate index terms (cell, cell+1) are formed once, (q * prime)
// for hashed levels, (q * stride) for dense ones, and the 2^D corners combine them with xor resp. add; `& mask` and one
// conditional subtract implement the modulo for every level class except LV_GENERIC, which only a pathological geometry
// produces: the stand-alone operator compiles the slow path in (GENERIC = true), the fused kernels refuse such a grid on the
// host (has_generic_level).
template <int D, int C, bool GENERIC = false>
__device__ __forceinline__ void gather_corners_generic(const float* __restrict__ table, const LevelGeom& g,
                                                       const uint32_t (&cell)[D], float (&v)[1 << D][C]) {
    const bool hashed = (g.flags & LV_HASHED) != 0;
    uint32_t term[D][2];
    term[0][0] = cell[0];
    term[0][1] = cell[0] + 1u;
    if (D > 1) {
        const uint32_t m1 = hashed ? 2654435761u : g.s1;
        term[1][0] = cell[1] * m1;
        term[1][1] = term[1][0] + m1;
    }
    if (D > 2) {
        const uint32_t m2 = hashed ? 805459861u : g.s2;
        term[2][0] = cell[2] * m2;
        term[2][1] = term[2][0] + m2;
    }
#pragma unroll
    for (int corner = 0; corner < (1 << D); ++corner) {
        uint32_t x = term[0][corner & 1], a = x;
#pragma unroll
        for (int d = 1; d < D; ++d) {
            const uint32_t t = term[d][(corner >> d) & 1];
            x ^= t;
            a += t;
        }
        uint32_t idx = (hashed ? x : a) & g.mask;
        idx = idx >= g.rows ? idx - g.rows : idx;
        if (GENERIC) {
            if ((g.flags & LV_GENERIC) != 0) idx = (hashed ? x : a) % g.rows;
        }
        load_row<C>(table + (size_t)(g.row0 + idx) * C, v[corner]);
    }
}

// Fast form of the fused kernels (D = 3, levels classified by make_grid_geom): the BYTE offsets of the eight corner rows from
// the table base, then eight loads with the table pointer in SGPRs and the 32-bit offset in a VGPR (global_load ... saddr) --
// no 64-bit address arithmetic per corner.  A lane takes the path of ITS level (the lanes of a wave may hold different levels; a
// path no lane needs is skipped; only offsets are formed inside the branches, so the loads stay one vector instruction per row):
//   LV_FASTDENSE  offset of corner (0,0,0) by a shift-add and two 24-bit multiply-adds with the level's byte strides, the
//                 other corners by seven adds.  The only modulo a dense level ever needs is at its upper boundary (cell + 1 ==
//                 resolution, i.e. a coordinate of exactly 1.0): such a lane takes the generic path;
//   LV_FASTHASH   power-of-two hashed level: the four (y, z) prime-term combinations once, then xor / and / shift-add per corner;
//   otherwise     the generic index arithmetic.
template <int C>
__device__ __forceinline__ void corner_offsets(const LevelGeom& g, const uint32_t (&cell)[3], uint32_t (&off)[8]) {
    constexpr uint32_t B = C * 4;
    // every corner row inside the level <=> the last one is; all byte offsets then stay below 2^31 (no wrap-around), and a
    // garbage cell of an out-of-range point -- whose result is discarded -- cannot address outside the table either
    uint32_t o00 = __umul24(cell[1], g.s1B) + (cell[0] * B + g.row0B);
    o00 = __umul24(cell[2], g.s2B) + o00;
    if ((g.flags & LV_FASTDENSE) && o00 < g.limB) {
        const uint32_t o10 = o00 + g.s1B, o01 = o00 + g.s2B, o11 = o01 + g.s1B;
        off[0] = o00; off[1] = o00 + B; off[2] = o10; off[3] = o10 + B;
        off[4] = o01; off[5] = o01 + B; off[6] = o11; off[7] = o11 + B;
    } else if (g.flags & LV_FASTHASH) {
        const uint32_t y0 = cell[1] * 2654435761u, y1 = y0 + 2654435761u;
        const uint32_t z0 = cell[2] * 805459861u, z1 = z0 + 805459861u;
        const uint32_t yz[4] = {y0 ^ z0, y1 ^ z0, y0 ^ z1, y1 ^ z1};
        const uint32_t x0 = cell[0], x1 = cell[0] + 1u;
#pragma unroll
        for (int corner = 0; corner < 8; ++corner)
            off[corner] = ((((corner & 1) ? x1 : x0) ^ yz[corner >> 1]) & g.mask) * B + g.row0B;
    } else {
        const bool hashed = (g.flags & LV_HASHED) != 0;
        const uint32_t m1 = hashed ? 2654435761u : g.s1, m2 = hashed ? 805459861u : g.s2;
        const uint32_t t0[2] = {cell[0], cell[0] + 1u}, t1[2] = {cell[1] * m1, cell[1] * m1 + m1}, t2[2] = {cell[2] * m2, cell[2] * m2 + m2};
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const uint32_t a0 = t0[corner & 1], a1 = t1[(corner >> 1) & 1], a2 = t2[corner >> 2];
            uint32_t idx = (hashed ? (a0 ^ a1 ^ a2) : (a0 + a1 + a2)) & g.mask;
            idx = idx >= g.rows ? idx - g.rows : idx;
            idx = idx < g.rows ? idx : g.rows - 1u;        // (garbage cell of a far out-of-range point: stay inside the level)
            off[corner] = idx * B + g.row0B;
        }
    }
}

// FAST = true: corner_offsets (per-lane branches on the level class).  Measured on MI355X, same box (profiles/r03_ab_experiments
// .txt): it removes 17 % of the sampler's VALU instructions and 11 us (210 -> 199) of its time, but in the kernels that live on
// memory-level parallelism (colour forward: HBM gather) or run two networks' worth of barriers (SDF forward / backward) the branch
// regions keep the compiler from hoisting the next level's loads above the current level's arithmetic, and those got 1-4 us
// SLOWER with 3-15 % fewer VALU instructions -- so only the sampler-side callers ask for it.
template <int D, int C, bool GENERIC = false, bool FAST = false>
__device__ __forceinline__ void gather_corners(const float* __restrict__ table, const LevelGeom& g,
                                               const uint32_t (&cell)[D], float (&v)[1 << D][C]) {
    if constexpr (D != 3 || GENERIC || !FAST) {
        gather_corners_generic<D, C, GENERIC>(table, g, cell, v);
    } else {
        uint32_t off[8];
        corner_offsets<C>(g, cell, off);
#ifdef NSA_X_GATHER_HOT      // timing-only ablation (WRONG numbers; tagged builds): every corner row comes from the table's first 64 KiB, i.e. from
#pragma unroll               // L1 / L2 -- the bound of what any prefetch of the sampler's gathers could give (profiles/r06_ab_experiments.txt r6s)
        for (int corner = 0; corner < 8; ++corner) off[corner] &= 0xFFF0u;
#endif
#pragma unroll
        for (int corner = 0; corner < 8; ++corner)
            load_row<C>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(table) + (size_t)off[corner]), v[corner]);
    }
}

// Smoothstep-weighted multilinear blend, corner order/product order as kernel_grid (:203-229).
template <int D, int C>
__device__ __forceinline__ void blend(const float (&v)[1 << D][C], const float (&w)[D], float (&out)[C]) {
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = 0.0f;
    float wt[1 << D];
    if constexpr (D == 3) {       // ((wx wy) wz), the association of the reference's running product, with the xy products shared
        const float nx = 1.0f - w[0], ny = 1.0f - w[1], nz = 1.0f - w[2];
        const float xy[4] = {nx * ny, w[0] * ny, nx * w[1], w[0] * w[1]};
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) wt[corner] = xy[corner & 3] * ((corner & 4) ? w[2] : nz);
    } else {
#pragma unroll
        for (int corner = 0; corner < (1 << D); ++corner) {
            wt[corner] = 1.0f;
#pragma unroll
            for (int d = 0; d < D; ++d) wt[corner] *= ((corner >> d) & 1) ? w[d] : 1.0f - w[d];
        }
    }
#pragma unroll
    for (int corner = 0; corner < (1 << D); ++corner)
#pragma unroll
        for (int c = 0; c < C; ++c) out[c] += wt[corner] * v[corner][c];
}

// Jacobian row d out/d x[gd] from the SAME corner values (kernel_grid :239-282 re-gathers them).
template <int D, int C>
__device__ __forceinline__ void jacobian_row(const float (&v)[1 << D][C], const float (&w)[D], const float (&dw)[D],
                                             float scale, int gd, float (&j)[C]) {
#pragma unroll
    for (int c = 0; c < C; ++c) j[c] = 0.0f;
#pragma unroll
    for (int face = 0; face < (1 << (D - 1)); ++face) {
        float wt = scale;
        int lo = 0;
#pragma unroll
        for (int nd = 0; nd < D - 1; ++nd) {
            const int d = (nd >= gd) ? nd + 1 : nd;
            if ((face >> nd) & 1) { wt *= w[d]; lo |= 1 << d; }
            else                  { wt *= 1.0f - w[d]; }
        }
        const int hi = lo | (1 << gd);
#pragma unroll
        for (int c = 0; c < C; ++c) j[c] += wt * (v[hi][c] - v[lo][c]) * dw[gd];
    }
}

}  // namespace nsa
