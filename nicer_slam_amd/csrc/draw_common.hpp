// draw_common.hpp -- the engine's counter-based generator (Philox4x32-10) and the body of the draw kernel, shared by k_draw
// (render_sampler.hip: nsa_draw) and k_track_begin_draw (track_tail.hip: the tracker's head launch, which makes the iteration's
// draws beside the ray lifting instead of in a graph node of their own).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace nsa {

// ---- all random draws of one sampler call in ONE launch, from a counter-based generator owned by the caller ------------------
// Philox4x32-10 (Salmon et al., SC'11 -- the generator behind torch.rand on the device): key = seed, counter = (index, region,
// call number).  `state` (device, 4 x uint64: seed, call number, ticket, unused) is advanced by the kernel itself -- the last
// workgroup to finish bumps the call number -- so a captured hipGraph draws fresh numbers on every replay without any host-side
// generator bookkeeping (torch's graph-safe generator costs two fills and two copies in front of every replay).
//   t_rand[n_rand]  uniforms in [0,1) (24 bits), region 0: the stratified jitter of ray_sampler.py:57-58
//   extra_idx       as k_draw_picks, from E keys of region 1                  (ray_sampler.py:148)
//   eik_idx[R]      floor(u S) from region 2                                  (ray_sampler.py:158)
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1;
        c[3] = (uint32_t)p0;
        c[0] = n0;
        c[2] = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }   // 2^-24

struct DrawArgs {
    unsigned long long* state;
    float* t_rand; uint64_t n_rand;
    uint32_t E, n_extra, R, S, rand_blocks;
    int32_t* extra_idx; int32_t* eik_idx;
};

__device__ __forceinline__ void draw4(const DrawArgs& a, uint32_t idx, uint32_t region, uint64_t call, uint64_t seed, float (&u)[4]) {
    uint32_t c[4] = {idx, region, (uint32_t)call, (uint32_t)(call >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = u01(c[i]);
}

// one workgroup (256 threads) of the draw launch: block `bid` of `nblocks` (rand blocks first, then pick blocks)
__device__ __forceinline__ void draw_block(const DrawArgs& a, const uint32_t bid, const uint32_t nblocks) {
    __shared__ __attribute__((aligned(16))) float key[1024];
    __shared__ uint32_t part[4][64];
    const uint64_t seed = a.state[0], call = a.state[1];
    if (bid < a.rand_blocks) {
        // (few workgroups, each looping: the closing ticket is one same-address device-scope atomic per workgroup, ~15 ns apiece)
        for (uint64_t i4 = (uint64_t)bid * 256 + threadIdx.x; i4 * 4 < a.n_rand; i4 += (uint64_t)a.rand_blocks * 256) {
            float u[4];                                                         // group of four consecutive draws
            draw4(a, (uint32_t)i4, 0u, call, seed, u);
            if (i4 * 4 + 3 < a.n_rand) *reinterpret_cast<float4*>(a.t_rand + i4 * 4) = make_float4(u[0], u[1], u[2], u[3]);
            else for (uint64_t k = 0; i4 * 4 + k < a.n_rand; ++k) a.t_rand[i4 * 4 + k] = u[k];
        }
    } else {
        // the picks: 16 keys per workgroup (the E^2 comparisons spread over E/16 workgroups), every workgroup regenerates all E
        // keys (E / 4 generator calls); thread (key k, segment s) = k + 16 s counts one sixteenth of the comparison range
        const uint32_t pb = bid - a.rand_blocks, n_pb = nblocks - a.rand_blocks;
        const uint32_t lane = threadIdx.x & 63, q = threadIdx.x >> 6;
        {
            float u[4];
            draw4(a, threadIdx.x, 1u, call, seed, u);
#pragma unroll
            for (int k = 0; k < 4; ++k) key[4 * threadIdx.x + k] = 4 * threadIdx.x + k < a.E ? u[k] : 2.0f;
        }
        __syncthreads();
        if (a.extra_idx) {
            const uint32_t kk = threadIdx.x & 15, sg = threadIdx.x >> 4;
            const uint32_t t = pb * 16 + kk;
            const float mine = key[t < 1024 ? t : 1023];
            uint32_t rank = 0;
            const float4* k4 = reinterpret_cast<const float4*>(key);
            const uint32_t n4 = (a.E + 3) / 4, per = (n4 + 15) / 16;
            const uint32_t lo = sg * per, hi = lo + per < n4 ? lo + per : n4;
            for (uint32_t j4 = lo; j4 < hi; ++j4) {
                const float4 o = k4[j4];
                const uint32_t j = 4 * j4;
                rank += (o.x < mine || (o.x == mine && j < t)) ? 1u : 0u;
                rank += (o.y < mine || (o.y == mine && j + 1 < t)) ? 1u : 0u;
                rank += (o.z < mine || (o.z == mine && j + 2 < t)) ? 1u : 0u;
                rank += (o.w < mine || (o.w == mine && j + 3 < t)) ? 1u : 0u;
            }
            rank += __shfl_xor(rank, 16);
            rank += __shfl_xor(rank, 32);
            if (lane < 16) part[q][lane] = rank;
            __syncthreads();
            if (threadIdx.x < 16 && t < a.E) {
                const uint32_t r = part[0][kk] + part[1][kk] + part[2][kk] + part[3][kk];
                if (r < a.n_extra) a.extra_idx[r] = (int32_t)t;
            }
        }
        if (a.eik_idx)
            for (uint32_t g = pb * 256 + threadIdx.x; g * 4 < a.R; g += n_pb * 256) {
                float u[4];
                draw4(a, g, 2u, call, seed, u);
                for (uint32_t k = 0; k < 4 && g * 4 + k < a.R; ++k) {
                    const uint32_t v = (uint32_t)(u[k] * (float)a.S);
                    a.eik_idx[g * 4 + k] = (int32_t)(v < a.S ? v : a.S - 1);
                }
            }
    }
    // every workgroup has read the call number by now; the last one to get here starts the next call
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* ticket = reinterpret_cast<unsigned*>(a.state + 2);
        if (atomicAdd(ticket, 1u) == nblocks - 1) {
            *ticket = 0u;
            a.state[1] = call + 1;
        }
    }
}

// launch shape of a draw: rand blocks (each loops) + pick blocks; fills `a`, returns the total number of workgroups (0: nothing to do)
inline uint32_t draw_launch_shape(DrawArgs& a, unsigned long long* state, uint64_t n_rand, float* t_rand, uint32_t E, uint32_t n_extra,
                                  uint32_t R, uint32_t S, int32_t* extra_idx, int32_t* eik_idx) {
    a.state = state;
    a.t_rand = t_rand; a.n_rand = n_rand; a.E = extra_idx ? E : 0; a.n_extra = extra_idx ? n_extra : 0; a.R = eik_idx ? R : 0; a.S = S;
    a.rand_blocks = (uint32_t)((n_rand + 1023) / 1024);
    if (a.rand_blocks > 128) a.rand_blocks = 128;
    a.extra_idx = extra_idx; a.eik_idx = eik_idx;
    uint32_t pick_blocks = 0;
    if (extra_idx) pick_blocks = (E + 15) / 16;
    else if (eik_idx) pick_blocks = 1;
    return a.rand_blocks + pick_blocks;
}
inline bool draw_args_ok(const void* state, uint64_t n_rand, const float* t_rand, uint32_t E, uint32_t n_extra, uint32_t R, uint32_t S,
                         const int32_t* extra_idx, const int32_t* eik_idx) {
    return !(!state || (n_rand && !t_rand) || n_rand > (1ull << 33) || (extra_idx && (E == 0 || E > 1024 || n_extra > E)) ||
             (eik_idx && (S == 0 || R == 0)));
}

}  // namespace nsa
