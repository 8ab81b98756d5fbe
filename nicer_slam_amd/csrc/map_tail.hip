// map_tail.hip -- the two remaining per-iteration passes of a mapping step that are pure HBM streaming:
//   k_update_voxels   the 64^3 visit counter fed by every sample of the batch
//   k_adam_table      torch.optim.Adam over a (large) parameter tensor in one pass
// Reference: SLAMNetwork.update_voxels (code/model/network.py:62-76); torch.optim.Adam as configured by
// code/training/volsdf_train.py:174 (betas (0.9, 0.99), eps 1e-15, no weight decay, no amsgrad).
#include "sdf_net.hpp"

namespace nsa {

// One lane per sample; consecutive samples of a ray fall into the same voxel in runs, which scatter_runs merges into
// one atomic per run.  Counts are exact in fp32 up to 2^24 visits per voxel (the reference counts in fp32 too).
__global__ __launch_bounds__(256) void k_update_voxels(PointSrc src, float* __restrict__ voxels, uint32_t res) {
    const uint32_t pid = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t key = 0xFFFFFFFFu;
    if (pid < src.P) {
        float x[3], z;
        uint32_t ray;
        load_point(src, pid, x, ray, z);
        const bool skip = fabsf(x[0]) > 0.99f || fabsf(x[1]) > 0.99f || fabsf(x[2]) > 0.99f;   // NaN: not skipped, as torch
        if (!skip) {
            long long q[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) q[d] = (long long)((x[d] + 1.0f) / 2.0f * (float)res);
            const long long flat = q[0] * res * res + q[1] * res + q[2];
            if (flat >= 0 && flat < (long long)res * res * res) key = (uint32_t)flat;
        }
    }
    float one[1] = {1.0f};
    scatter_runs<1>(voxels, key, one, lane);
}

// 30-bit Morton code of each point's cell in a 1024^3 lattice over [-1,1]^3 (points outside are clamped): sorting the
// points by it makes the lanes of a wave spatial neighbours, so their grid gathers share cache lines and the table-gradient
// scatter merges whole runs of equal rows (grid_common.hpp::scatter_runs).
__device__ __forceinline__ uint32_t spread10(uint32_t v) {
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void k_morton_keys(PointSrc src, int32_t* __restrict__ keys) {
    const uint32_t pid = blockIdx.x * 256 + threadIdx.x;
    if (pid >= src.P) return;
    float x[3], z;
    uint32_t ray;
    load_point(src, pid, x, ray, z);
    uint32_t c[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float u = fminf(fmaxf((x[d] + 1.0f) * 512.0f, 0.0f), 1023.0f);     // NaN -> 0
        c[d] = (uint32_t)u;
    }
    keys[pid] = (int32_t)(spread10(c[0]) | (spread10(c[1]) << 1) | (spread10(c[2]) << 2));
}

struct AdamTableArgs {
    float* p; const float* g; float* m; float* v;
    uint64_t n;
    float w1;          // 1 - beta1
    float beta2, w2;   // beta2, 1 - beta2
    float step_size;   // lr / (1 - beta1^t)
    float bc2_sqrt;    // sqrt(1 - beta2^t)
    float eps;
};

// Same operation order as torch's foreach Adam: m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, 1-b2);
// denom = sqrt(v) / bc2_sqrt + eps; p.addcdiv_(m, denom, -step_size).
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamTableArgs& a) {
    m = m + a.w1 * (g - m);
    v = v * a.beta2 + a.w2 * g * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p - a.step_size * (m / denom);
}

// (A variant with 2 / 4 float4 groups per thread and all their loads issued first measured the same 4.9 TB/s = 0.78 of the 6.29 TB/s
// streaming-copy rate on a 1 GiB table, profiles/r04_ab_experiments.txt r4h: seven concurrent streams, not load depth, set the rate.)
__global__ __launch_bounds__(256) void k_adam_table(AdamTableArgs a) {
    const uint64_t n4 = a.n / 4;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 p = reinterpret_cast<float4*>(a.p)[i];
        const float4 g = reinterpret_cast<const float4*>(a.g)[i];
        float4 m = reinterpret_cast<float4*>(a.m)[i];
        float4 v = reinterpret_cast<float4*>(a.v)[i];
        adam_one(p.x, g.x, m.x, v.x, a);
        adam_one(p.y, g.y, m.y, v.y, a);
        adam_one(p.z, g.z, m.z, v.z, a);
        adam_one(p.w, g.w, m.w, v.w, a);
        reinterpret_cast<float4*>(a.p)[i] = p;
        reinterpret_cast<float4*>(a.m)[i] = m;
        reinterpret_cast<float4*>(a.v)[i] = v;
    }
    if (blockIdx.x == 0) {
        const uint64_t i = n4 * 4 + threadIdx.x;
        if (i < a.n) adam_one(a.p[i], a.g[i], a.m[i], a.v[i], a);
    }
}

// ---- packed MLP parameter blocks in one launch (fused/pack.py::pack_blocks) ----------------------------------------------------
// out[o] = word perm[o] of  concat( split3(flat[ia]) viewed as [group][piece][lane][4 x (2 bf16)],  flat[iv] ):
// the gather plan of pack.py (index of every fp32 weight an MFMA lane holds, of every per-feature value, and the block order)
// applied with the exact 3-way bf16 split of the weights -- hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), each
// round-to-nearest-even like torch's .to(torch.bfloat16).  One launch instead of the gather / 3 conversions / 2 subtractions /
// stack / cat / gather that torch makes of it (a mapping iteration re-packs three networks: its MLPs move every step).
__device__ __forceinline__ uint32_t bf16_rne(float x) {
    const uint32_t u = __float_as_uint(x);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;          // NaN stays NaN (quiet)
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ uint32_t bf16_piece(float x, int piece) {
    const uint32_t hi = bf16_rne(x);
    if (piece == 0) return hi;
    const float r = x - __uint_as_float(hi << 16);
    const uint32_t mid = bf16_rne(r);
    if (piece == 1) return mid;
    return bf16_rne(r - __uint_as_float(mid << 16));
}

__global__ __launch_bounds__(256) void k_pack_blocks(const float* __restrict__ flat, const int64_t* __restrict__ ia,
                                                     uint64_t n_a_words, const int64_t* __restrict__ iv,
                                                     const int64_t* __restrict__ perm, uint64_t n_out, uint32_t* __restrict__ out) {
    const uint64_t o = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    const uint64_t p = (uint64_t)perm[o];
    if (p < n_a_words) {                                   // [group][piece][lane][word w = bf16 elements 2w, 2w+1]
        const uint64_t g = p / 768;
        const uint32_t rem = (uint32_t)(p - g * 768);
        const int piece = rem >> 8, lane = (rem & 255) >> 2, w = rem & 3;
        const int64_t* src = ia + (g * 64 + lane) * 8 + 2 * w;
        const float x0 = flat[src[0]], x1 = flat[src[1]];
        out[o] = bf16_piece(x0, piece) | (bf16_piece(x1, piece) << 16);
    } else {
        out[o] = __float_as_uint(flat[iv[p - n_a_words]]);
    }
}

}  // namespace nsa

extern "C" {

int nsa_pack_blocks(const float* flat, const int64_t* a_index, uint64_t n_a, const int64_t* v_index, uint64_t n_v,
                    const int64_t* order, uint64_t n_out, float* out, nsa_stream_t stream) {
    using namespace nsa;
    if (n_out == 0) return NSA_OK;
    if (!flat || !order || !out || (n_a && !a_index) || (n_v && !v_index) || (n_a % 512) != 0) return NSA_EBADARG;
    const uint64_t n_a_words = n_a / 8 * 3 * 4;            // 8 weights of a lane -> 3 pieces x 4 words
    if (n_out > n_a_words + n_v || n_out > 0x7FFFFFFFull * 256) return NSA_EBADARG;
    launch_begin();
    hipLaunchKernelGGL(k_pack_blocks, dim3((uint32_t)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flat, a_index,
                       n_a_words, v_index, order, n_out, reinterpret_cast<uint32_t*>(out));
    return launch_end();
}

int nsa_update_voxels(const nsa_points_t* pts, float* voxels, uint32_t res, nsa_stream_t stream) {
    using namespace nsa;
    if (!pts || !voxels || res == 0 || res > 1024) return NSA_EBADARG;
    if (pts->P == 0) return NSA_OK;
    if (!pts->points && (!pts->rays_o || !pts->rays_d || !pts->z_vals || pts->S == 0)) return NSA_EBADARG;
    const PointSrc src{pts->rays_o, pts->rays_d, pts->z_vals, pts->points, pts->P, pts->S, nullptr};
    launch_begin();
    hipLaunchKernelGGL(k_update_voxels, dim3((pts->P + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, voxels, res);
    return launch_end();
}

int nsa_morton_keys(const nsa_points_t* pts, int32_t* keys, nsa_stream_t stream) {
    using namespace nsa;
    if (!pts || !keys) return NSA_EBADARG;
    if (pts->P == 0) return NSA_OK;
    if (!pts->points && (!pts->rays_o || !pts->rays_d || !pts->z_vals || pts->S == 0)) return NSA_EBADARG;
    const PointSrc src{pts->rays_o, pts->rays_d, pts->z_vals, pts->points, pts->P, pts->S, nullptr};
    launch_begin();
    hipLaunchKernelGGL(k_morton_keys, dim3((pts->P + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, keys);
    return launch_end();
}

int nsa_adam_table_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, uint32_t step,
                        float lr, float beta1, float beta2, float eps, nsa_stream_t stream) {
    using namespace nsa;
    if (!param || !grad || !exp_avg || !exp_avg_sq || step == 0) return NSA_EBADARG;
    if (n == 0) return NSA_OK;
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15u) return NSA_EBADARG;            // float4 path
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamTableArgs a{param, grad, exp_avg, exp_avg_sq, n, 1.0f - beta1, beta2, 1.0f - beta2,
                    (float)((double)lr / bc1), (float)sqrt(bc2), eps};
    const uint64_t n4 = n / 4;
    uint64_t blocks = (n4 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;      // grid-stride: 32 blocks per CU keeps every HBM channel busy
    if (blocks == 0) blocks = 1;
    launch_begin();
    hipLaunchKernelGGL(k_adam_table, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return launch_end();
}

}  // extern "C"
