// map_tail.hip -- the two remaining per-iteration passes of a mapping step that are pure HBM streaming:
//   k_update_voxels   the 64^3 visit counter fed by every sample of the batch
//   k_adam_table      torch.optim.Adam over a (large) parameter tensor in one pass
// Reference: SLAMNetwork.update_voxels (code/model/network.py:62-76); torch.optim.Adam as configured by
// code/training/volsdf_train.py:174 (betas (0.9, 0.99), eps 1e-15, no weight decay, no amsgrad).
#include <cstdlib>
#include <cstring>
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

// ---- argsort of the Morton keys: stable LSD radix sort, 8-bit digits, two launches per pass ------------------------------------
// (replaces torch.sort = rocprim's block sort + ~26 merge launches per call.)  A pass: k_radix_hist counts the digits of every
// block's tile into counts[block][256]; k_radix_scatter turns them into the block's output offsets itself (digit totals over all
// blocks + the earlier blocks' share: nb x 1 KiB of L2 reads per block, nb <= 256, so no scan launch in between), then ranks its
// keys stably -- a wave owns a contiguous run of its block's tile and walks it 64 keys at a time; the lanes holding equal digits
// find each other with 8 ballots, the lowest of them advances the wave's running offset of that digit in LDS.  No inter-block
// waiting, no atomics on global memory: the result is deterministic (equal keys keep their index order).
constexpr uint32_t RS_WAVES = 4;
struct RadixArgs {
    const uint32_t* keys_in;
    const uint32_t* vals_in;      // nullptr: the identity (first pass)
    uint32_t* keys_out;           // nullptr: keys not needed any more (last pass)
    uint32_t* vals_out;
    uint32_t* counts;             // [nb][256]
    uint32_t P, nb, per_wave, shift;
};

constexpr int RS_CHUNK = 16;      // rounds (of 64 keys) a wave loads before it ranks them: the loads of a chunk are in flight together

// keys (and, with `vals`, their payloads) of one chunk of this wave's run; slots past the run / past P hold `valid` = false
struct RadixChunk {
    uint32_t key[RS_CHUNK], val[RS_CHUNK];
    bool valid[RS_CHUNK];
};
__device__ __forceinline__ void radix_load(const RadixArgs& a, uint64_t start, uint32_t r0, uint32_t lane, bool vals, RadixChunk& c) {
#pragma unroll
    for (int j = 0; j < RS_CHUNK; ++j) {
        const uint32_t r = r0 + 64u * j;
        const uint64_t e = start + r + lane;
        c.valid[j] = r < a.per_wave && e < a.P;
        c.key[j] = c.valid[j] ? a.keys_in[e] : 0u;
        c.val[j] = (uint32_t)e;
        if (vals && a.vals_in) c.val[j] = c.valid[j] ? a.vals_in[e] : 0u;
    }
}

__global__ __launch_bounds__(256) void k_radix_hist(RadixArgs a) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t start = ((uint64_t)blockIdx.x * RS_WAVES + w) * a.per_wave;
    for (uint32_t r0 = 0; r0 < a.per_wave; r0 += 64u * RS_CHUNK) {
        RadixChunk c;
        radix_load(a, start, r0, lane, false, c);
#pragma unroll
        for (int j = 0; j < RS_CHUNK; ++j)
            if (c.valid[j]) atomicAdd(&h[(c.key[j] >> a.shift) & 255u], 1u);
    }
    __syncthreads();
    a.counts[blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_radix_scatter(RadixArgs a) {
    __shared__ uint32_t wh[RS_WAVES][256];
    __shared__ uint32_t run[RS_WAVES][256];
    __shared__ uint32_t scan[256];
    const uint32_t t = threadIdx.x, w = t >> 6, lane = t & 63;
    // digit t: keys of all blocks, and of the blocks before this one
    uint32_t total = 0, before = 0;
#pragma unroll 8
    for (uint32_t b = 0; b < a.nb; ++b) {
        const uint32_t c = a.counts[b * 256 + t];
        total += c;
        before += b < blockIdx.x ? c : 0u;
    }
    scan[t] = total;
#pragma unroll
    for (uint32_t wv = 0; wv < RS_WAVES; ++wv) wh[wv][t] = 0;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {             // inclusive scan over the digits
        const uint32_t v = t >= off ? scan[t - off] : 0u;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    const uint64_t start = ((uint64_t)blockIdx.x * RS_WAVES + w) * a.per_wave;
    const bool single = a.per_wave <= 64u * RS_CHUNK;          // the whole run in one chunk: loaded once, kept in registers
    RadixChunk c;
    for (uint32_t r0 = 0; r0 < a.per_wave; r0 += 64u * RS_CHUNK) {      // digits of this wave's run
        radix_load(a, start, r0, lane, true, c);
#pragma unroll
        for (int j = 0; j < RS_CHUNK; ++j)
            if (c.valid[j]) atomicAdd(&wh[w][(c.key[j] >> a.shift) & 255u], 1u);
    }
    __syncthreads();
    {
        uint32_t acc = scan[t] - total + before;               // first output slot of digit t for this block
#pragma unroll
        for (uint32_t wv = 0; wv < RS_WAVES; ++wv) {
            run[wv][t] = acc;
            acc += wh[wv][t];
        }
    }
    __syncthreads();
    for (uint32_t r0 = 0; r0 < a.per_wave; r0 += 64u * RS_CHUNK) {
        if (!single) radix_load(a, start, r0, lane, true, c);
#pragma unroll
        for (int j = 0; j < RS_CHUNK; ++j) {
            const bool valid = c.valid[j];
            const uint32_t d = (c.key[j] >> a.shift) & 255u;
            unsigned long long same = __ballot(valid);         // valid lanes holding this lane's digit
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long bal = __ballot(bit);
                same &= bit ? bal : ~bal;
            }
            const uint32_t rank = __popcll(same & ((1ull << lane) - 1ull)), cnt = __popcll(same);
            const uint32_t base = valid ? run[w][d] : 0u;
            __builtin_amdgcn_wave_barrier();
            if (valid && rank == 0) run[w][d] = base + cnt;    // (same wave, program order: the next round reads it)
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                const uint32_t pos = base + rank;
                if (a.keys_out) a.keys_out[pos] = c.key[j];
                a.vals_out[pos] = c.val[j];
            }
        }
    }
}

struct AdamTableArgs {
    float* p; float* g; float* m; float* v;
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

// Block orders measured for the dense step over a 1 GiB table (profiles/r04_ab_experiments.txt r4h, r05 r5E, tools/ab_adam.py of those
// rounds): a grid capped at 32 blocks per CU that strides through the buffer reached 4.9 TB/s; one block per contiguous 4 KiB chunk,
// front to back, 5.7; the same with non-temporal accesses 6.0-6.4.  The front-to-back forms are what is left: k_adam_table_linear
// (tables that fit the MALL, and the CLEAR form) and k_adam_table_linear_nt (tables beyond it).
// CLEAR: the gradient is consumed -- every element read is left zero, so that a persistent gradient buffer needs no separate
// zero fill before the next backward pass scatters into it (fused/tablegrad.py; the reference's zero_grad + dense autograd
// gradient, volsdf_train.py:547-576 and hashgrid.py:117-118, as one pass).  Untouched 16-byte groups are not rewritten.
// One block = one contiguous chunk (4 KiB per float4 group), no loop: the dispatcher walks the buffer front to back, so HBM sees one linear write stream (6.8 TB/s on a
// 1 GiB buffer; a grid capped at 8 blocks per CU that strides through the buffer reached 4.8 -- tools/micro/fill_bench.py, r5A).
// (two / four float4 groups per thread: no faster, r5A; non-temporal stores: 3 % slower, r5E)
__global__ __launch_bounds__(256) void k_fill_zero(float* __restrict__ p, uint64_t n) {
    const uint64_t n4 = n / 4;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) reinterpret_cast<float4*>(p)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (blockIdx.x == 0 && n4 * 4 + threadIdx.x < n) p[n4 * 4 + threadIdx.x] = 0.0f;
}

// k_adam_table with the same front-to-back block order: block b owns float4 groups [b * 256 G, (b + 1) * 256 G)
__device__ __forceinline__ float4 nt_load4(const float4* p) {
    float4 r;
    r.x = __builtin_nontemporal_load(&p->x); r.y = __builtin_nontemporal_load(&p->y);
    r.z = __builtin_nontemporal_load(&p->z); r.w = __builtin_nontemporal_load(&p->w);
    return r;
}
__device__ __forceinline__ void nt_store4(float4* p, const float4& v) {
    __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
}

// The linear kernel with non-temporal accesses (gfx950: the `nt` bit on global_load / global_store_dwordx4 -- streamed lines are not
// kept in L2): 5-8 % faster again on the 1 GiB table (1355 -> 1265 us on the slower of two boxes, r5E; the hint on the loads only or
// on the stores only measured in between).
__global__ __launch_bounds__(256) void k_adam_table_linear_nt(AdamTableArgs a) {
    const uint64_t n4 = a.n / 4;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        float4 p = nt_load4(reinterpret_cast<const float4*>(a.p) + i);
        const float4 g = nt_load4(reinterpret_cast<const float4*>(a.g) + i);
        float4 m = nt_load4(reinterpret_cast<const float4*>(a.m) + i);
        float4 v = nt_load4(reinterpret_cast<const float4*>(a.v) + i);
        adam_one(p.x, g.x, m.x, v.x, a);
        adam_one(p.y, g.y, m.y, v.y, a);
        adam_one(p.z, g.z, m.z, v.z, a);
        adam_one(p.w, g.w, m.w, v.w, a);
        nt_store4(reinterpret_cast<float4*>(a.p) + i, p);
        nt_store4(reinterpret_cast<float4*>(a.m) + i, m);
        nt_store4(reinterpret_cast<float4*>(a.v) + i, v);
    }
    if (blockIdx.x == 0) {
        const uint64_t t = n4 * 4 + threadIdx.x;
        if (t < a.n) adam_one(a.p[t], a.g[t], a.m[t], a.v[t], a);
    }
}

template <bool CLEAR>
__global__ __launch_bounds__(256) void k_adam_table_linear(AdamTableArgs a) {
    const uint64_t n4 = a.n / 4;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
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
        if (CLEAR && (g.x != 0.0f || g.y != 0.0f || g.z != 0.0f || g.w != 0.0f))        // (NaN != 0: cleared too)
            reinterpret_cast<float4*>(a.g)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (blockIdx.x == 0) {
        const uint64_t t = n4 * 4 + threadIdx.x;
        if (t < a.n) {
            adam_one(a.p[t], a.g[t], a.m[t], a.v[t], a);
            if (CLEAR) a.g[t] = 0.0f;
        }
    }
}

// The step torch 1.11 (the reference's environment, env_yamls/nicer-slam.yaml:62) takes for a parameter that received NO gradient in
// this iteration: there optimizer.zero_grad() (volsdf_train.py:547) leaves a ZERO tensor, so Adam still decays both moments and moves the
// parameter along its momentum -- the fine table during stage "coarse", the colour table during color_stage "base" (:550-555).  Same
// arithmetic as adam_one with g = 0 (m + w1 (0 - m); v beta2 + w2 0 0), without reading a gradient: 3 reads + 3 writes per element
// instead of a zero fill + 4 reads + 3 writes.  NT: non-temporal accesses for tables beyond the MALL, as k_adam_table_linear_nt.
template <bool NT>
__global__ __launch_bounds__(256) void k_adam_table_zero_grad(AdamTableArgs a) {
    const uint64_t n4 = a.n / 4;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        float4 p, m, v;
        if (NT) {
            p = nt_load4(reinterpret_cast<const float4*>(a.p) + i);
            m = nt_load4(reinterpret_cast<const float4*>(a.m) + i);
            v = nt_load4(reinterpret_cast<const float4*>(a.v) + i);
        } else {
            p = reinterpret_cast<const float4*>(a.p)[i];
            m = reinterpret_cast<const float4*>(a.m)[i];
            v = reinterpret_cast<const float4*>(a.v)[i];
        }
        adam_one(p.x, 0.0f, m.x, v.x, a);
        adam_one(p.y, 0.0f, m.y, v.y, a);
        adam_one(p.z, 0.0f, m.z, v.z, a);
        adam_one(p.w, 0.0f, m.w, v.w, a);
        if (NT) {
            nt_store4(reinterpret_cast<float4*>(a.p) + i, p);
            nt_store4(reinterpret_cast<float4*>(a.m) + i, m);
            nt_store4(reinterpret_cast<float4*>(a.v) + i, v);
        } else {
            reinterpret_cast<float4*>(a.p)[i] = p;
            reinterpret_cast<float4*>(a.m)[i] = m;
            reinterpret_cast<float4*>(a.v)[i] = v;
        }
    }
    if (blockIdx.x == 0) {
        const uint64_t t = n4 * 4 + threadIdx.x;
        if (t < a.n) adam_one(a.p[t], 0.0f, a.m[t], a.v[t], a);
    }
}

__global__ __launch_bounds__(256) void k_adam_table_zero_grad_scalar(AdamTableArgs a) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) adam_one(a.p[i], 0.0f, a.m[i], a.v[i], a);
}

// any alignment (gradients that are views into a larger buffer, e.g. FlatWeightNorm's): one element per thread
template <bool CLEAR>
__global__ __launch_bounds__(256) void k_adam_table_scalar(AdamTableArgs a) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {
        adam_one(a.p[i], a.g[i], a.m[i], a.v[i], a);
        if (CLEAR) a.g[i] = 0.0f;
    }
}

// Adam over up to ADAM_MULTI_MAX small tensors in ONE launch (the ~15 weight_v / weight_g / bias tensors of the two trained MLPs:
// a launch each was 13 x ~4.5 us of device time per mapping iteration for a few thousand floats).  Block b works on 256 elements
// of segment seg_of(b); per-segment step sizes (lr and step count may differ between parameter groups).
constexpr int ADAM_MULTI_MAX = 24;
struct AdamMultiArgs {
    float* p[ADAM_MULTI_MAX]; float* g[ADAM_MULTI_MAX]; float* m[ADAM_MULTI_MAX]; float* v[ADAM_MULTI_MAX];
    uint32_t n[ADAM_MULTI_MAX];
    uint32_t block0[ADAM_MULTI_MAX + 1];      // first block of every segment
    float step_size[ADAM_MULTI_MAX], bc2_sqrt[ADAM_MULTI_MAX];
    float w1, beta2, w2, eps;
    uint32_t count;
};

__global__ __launch_bounds__(256) void k_adam_multi(AdamMultiArgs a) {
    uint32_t s = 0;
    while (s + 1 < a.count && blockIdx.x >= a.block0[s + 1]) ++s;
    const uint32_t i = (blockIdx.x - a.block0[s]) * 256 + threadIdx.x;
    if (i >= a.n[s]) return;
    AdamTableArgs one;
    one.w1 = a.w1; one.beta2 = a.beta2; one.w2 = a.w2; one.eps = a.eps;
    one.step_size = a.step_size[s]; one.bc2_sqrt = a.bc2_sqrt[s];
    adam_one(a.p[s][i], a.g[s][i], a.m[s][i], a.v[s][i], one);
}

// ---- weight-normed MLP parameters <-> the flat effective parameter vector, one launch per direction ---------------------------
// flat = [W_0 (rows x cols, row-major), b_0, W_1, b_1, .., 0] with W_l[r,:] = v_l[r,:] * g_l[r] / ||v_l[r,:]||  -- what
// torch._weight_norm(v, g, dim=0) computes per layer (nn.utils.weight_norm of code/model/base_networks.py:137-141, 376-379) followed
// by the reshape / cat of fused/pack.py::flat_params.  One wave per weight row; the block after the last row writes the trailing 0.
constexpr int WN_MAX_LAYERS = 8;
struct WnArgs {
    const float* v[WN_MAX_LAYERS];
    const float* g[WN_MAX_LAYERS];
    const float* bias[WN_MAX_LAYERS];
    uint32_t rows[WN_MAX_LAYERS], cols[WN_MAX_LAYERS];
    uint32_t row0[WN_MAX_LAYERS + 1];      // first global row of layer l; row0[n] = number of rows
    uint32_t off[WN_MAX_LAYERS + 1];       // first flat element of layer l; off[n] = index of the trailing zero
    uint32_t n;
    float* flat;                           // forward: out
    float* norms;                          // forward: out [row0[n]]; backward: in
    const float* g_flat;                   // backward: cotangent of flat
    float* g_params;                       // backward: out, per layer [g_v (rows x cols) | g_g (rows) | g_bias (rows)]
};

__device__ __forceinline__ float wave_sum(float s) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    return s;
}

__device__ __forceinline__ int wn_layer_of(const WnArgs& a, uint32_t row) {
    int l = 0;
    while (l + 1 < (int)a.n && row >= a.row0[l + 1]) ++l;
    return l;
}

__global__ __launch_bounds__(64) void k_weight_norm_flat(WnArgs a) {
    const uint32_t row = blockIdx.x, tid = threadIdx.x;
    if (row == a.row0[a.n]) {
        if (tid == 0) a.flat[a.off[a.n]] = 0.0f;
        return;
    }
    const int l = wn_layer_of(a, row);
    const uint32_t r = row - a.row0[l], cols = a.cols[l];
    const float* v = a.v[l] + (size_t)r * cols;
    float s = 0.0f;
    for (uint32_t c = tid; c < cols; c += 64) s = fmaf(v[c], v[c], s);
    const float norm = sqrtf(wave_sum(s));
    const float rnorm = 1.0f / norm, gr = a.g[l][r];
    float* w = a.flat + a.off[l] + (size_t)r * cols;
    for (uint32_t c = tid; c < cols; c += 64) w[c] = v[c] * gr * rnorm;      // operation order of ATen's weight_norm kernel
    if (tid == 0) {
        a.norms[row] = norm;
        a.flat[a.off[l] + (size_t)a.rows[l] * cols + r] = a.bias[l][r];
    }
}

// g_g[r] = <g_W[r,:], v[r,:]> / ||v||,  g_v[r,:] = g[r] (g_W[r,:] / ||v|| - v[r,:] <g_W[r,:], v[r,:]> / ||v||^3),  g_bias = its slice
// (the formulas of ATen's weight_norm backward, so that the composed and the fused engine differ by summation order only).
__global__ __launch_bounds__(64) void k_weight_norm_flat_bwd(WnArgs a) {
    const uint32_t row = blockIdx.x, tid = threadIdx.x;
    const int l = wn_layer_of(a, row);
    const uint32_t r = row - a.row0[l], cols = a.cols[l], rows = a.rows[l];
    const float* v = a.v[l] + (size_t)r * cols;
    const float* gw = a.g_flat + a.off[l] + (size_t)r * cols;
    float s = 0.0f;
    for (uint32_t c = tid; c < cols; c += 64) s = fmaf(gw[c], v[c], s);
    s = wave_sum(s);
    const float rnorm = 1.0f / a.norms[row], gr = a.g[l][r];
    const float rnorm3 = rnorm * rnorm * rnorm;
    float* out = a.g_params + a.off[l] + a.row0[l];                       // = sum over earlier layers of rows * (cols + 2)
    for (uint32_t c = tid; c < cols; c += 64) out[(size_t)r * cols + c] = gr * (rnorm * gw[c] - rnorm3 * v[c] * s);
    if (tid == 0) {
        out[(size_t)rows * cols + r] = s * rnorm;
        out[(size_t)rows * cols + rows + r] = a.g_flat[a.off[l] + (size_t)rows * cols + r];
    }
}

// dst[p] = src[order ? order[p] : p] for p < P, 0 up to n: one extra emission row (the per-point cotangent of the sdf value, in
// launch order) so that the last layer's sdf-row gradient is one more nsa_emit_gemm product instead of a library GEMV.
__global__ __launch_bounds__(256) void k_emit_row(float* __restrict__ dst, const float* __restrict__ src,
                                                  const int32_t* __restrict__ order, uint32_t P, uint64_t n, float fill) {
    const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    dst[p] = p < P ? (src ? src[order ? (uint32_t)order[p] : (uint32_t)p] : fill) : 0.0f;
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

// form 2 (mlp_common.hpp::NSA_FORM): the fragment triple is [fp16 h0 | fp16 h1 | bf16 round-to-nearest] of the weight --
// h0 = fp16(512 w), h1 = fp16(512 w - h0), both round-to-nearest (h0 + h1 = 512 w to 2^-23; |w| >= 127.97 becomes +-inf in h0 and
// the kernels' results non-finite: loud, not wrong); the third slot is what the bf16-operand kernels multiply with.
__device__ __forceinline__ uint32_t f16_bits(float x) {
    const _Float16 h = (_Float16)x;
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}
__device__ __forceinline__ uint32_t h2_piece(float x, int piece) {
    if (piece == 2) return bf16_rne(x);
    const float t = x * kWScale;
    const _Float16 h0 = (_Float16)t;
    if (piece == 0) return f16_bits(t);
    return f16_bits(t - (float)h0);
}
__device__ __forceinline__ uint32_t weight_piece(float x, int piece) {
    if constexpr (kForm == 2) return h2_piece(x, piece);
    else return bf16_piece(x, piece);
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
        out[o] = weight_piece(x0, piece) | (weight_piece(x1, piece) << 16);
    } else {
        out[o] = __float_as_uint(flat[iv[p - n_a_words]]);
    }
}

}  // namespace nsa

extern "C" {

int nsa_operand_form(void) { return nsa::kForm; }

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

static void radix_geometry(uint32_t P, uint32_t& nb, uint32_t& per_wave) {
    nb = (P + 4095u) / 4096u;
    if (nb < 1) nb = 1;
    if (nb > 256) nb = 256;
    const uint64_t waves = (uint64_t)nb * nsa::RS_WAVES;
    per_wave = (uint32_t)((((uint64_t)P + waves - 1) / waves + 63) / 64 * 64);
}

uint64_t nsa_morton_order_workspace(uint32_t P) {
    uint32_t nb, per_wave;
    radix_geometry(P, nb, per_wave);
    return 3ull * P + (uint64_t)nb * 256;
}

int nsa_morton_order(const nsa_points_t* pts, int32_t* order, uint32_t* workspace, uint32_t key_bits, nsa_stream_t stream) {
    using namespace nsa;
    if (!pts || !order || !workspace || key_bits < 1 || key_bits > 30) return NSA_EBADARG;
    const uint32_t P = pts->P;
    if (P == 0) return NSA_OK;
    if (P > 0x7FFFFFFFu) return NSA_EBADARG;
    uint32_t* keys[2] = {workspace, workspace + P};
    uint32_t* tmp = workspace + 2ull * P;
    uint32_t* counts = workspace + 3ull * P;
    const int rc = nsa_morton_keys(pts, reinterpret_cast<int32_t*>(keys[0]), stream);
    if (rc != NSA_OK) return rc;
    RadixArgs a;
    a.P = P;
    radix_geometry(P, a.nb, a.per_wave);
    a.counts = counts;
    const uint32_t passes = (key_bits + 7) / 8, shift0 = 30 - key_bits;
    launch_begin();
    for (uint32_t i = 0; i < passes; ++i) {
        uint32_t* v_out = ((passes - 1 - i) & 1u) ? tmp : reinterpret_cast<uint32_t*>(order);
        const uint32_t* v_in = i == 0 ? nullptr : (((passes - i) & 1u) ? tmp : reinterpret_cast<uint32_t*>(order));
        a.keys_in = keys[i & 1];
        a.keys_out = i + 1 < passes ? keys[(i + 1) & 1] : nullptr;
        a.vals_in = v_in;
        a.vals_out = v_out;
        a.shift = shift0 + 8 * i;
        hipLaunchKernelGGL(k_radix_hist, dim3(a.nb), dim3(256), 0, (hipStream_t)stream, a);
        hipLaunchKernelGGL(k_radix_scatter, dim3(a.nb), dim3(256), 0, (hipStream_t)stream, a);
    }
    return launch_end();
}

static int adam_table_launch(float* param, float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, uint32_t step,
                             float lr, float beta1, float beta2, float eps, bool clear, nsa_stream_t stream) {
    using namespace nsa;
    if (!param || !grad || !exp_avg || !exp_avg_sq || step == 0) return NSA_EBADARG;
    if (n == 0) return NSA_OK;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) |
                           reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq);
    if (bits & 3u) return NSA_EBADARG;
    const bool vec = (bits & 15u) == 0;                                                  // float4 path
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamTableArgs a{param, grad, exp_avg, exp_avg_sq, n, 1.0f - beta1, beta2, 1.0f - beta2,
                    (float)((double)lr / bc1), (float)sqrt(bc2), eps};
    const uint64_t n4 = n / 4;
    launch_begin();
    if (!vec) {
        uint64_t sb = (n + 255) / 256;
        if (sb > 256 * 32) sb = 256 * 32;
        if (clear) hipLaunchKernelGGL(k_adam_table_scalar<true>, dim3((uint32_t)sb), dim3(256), 0, (hipStream_t)stream, a);
        else       hipLaunchKernelGGL(k_adam_table_scalar<false>, dim3((uint32_t)sb), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        const uint64_t lb = (n4 + 255) / 256 ? (n4 + 255) / 256 : 1;
        if (lb > 0x7FFFFFFFull) return NSA_EBADARG;
        const dim3 grid((uint32_t)lb), block(256);
        // non-temporal only where nothing of the tensor could stay cached anyway: a table that fits the 256 MB of MALL (the SDF
        // tables, 4 and 36 MiB) is gathered from by the very next forward pass and should stay there
        if (!clear && n >= (1ull << 26)) hipLaunchKernelGGL(k_adam_table_linear_nt, grid, block, 0, (hipStream_t)stream, a);
        else if (clear)                  hipLaunchKernelGGL(k_adam_table_linear<true>, grid, block, 0, (hipStream_t)stream, a);
        else                             hipLaunchKernelGGL(k_adam_table_linear<false>, grid, block, 0, (hipStream_t)stream, a);
    }
    return launch_end();
}

int nsa_adam_table_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, uint32_t step,
                        float lr, float beta1, float beta2, float eps, nsa_stream_t stream) {
    return adam_table_launch(param, const_cast<float*>(grad), exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, false, stream);
}

int nsa_adam_table_step_clear(float* param, float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, uint32_t step,
                              float lr, float beta1, float beta2, float eps, nsa_stream_t stream) {
    return adam_table_launch(param, grad, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, true, stream);
}

int nsa_adam_table_step_zero_grad(float* param, float* exp_avg, float* exp_avg_sq, uint64_t n, uint32_t step, float lr, float beta1,
                                  float beta2, float eps, nsa_stream_t stream) {
    using namespace nsa;
    if (!param || !exp_avg || !exp_avg_sq || step == 0) return NSA_EBADARG;
    if (n == 0) return NSA_OK;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq);
    if (bits & 3u) return NSA_EBADARG;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamTableArgs a{param, nullptr, exp_avg, exp_avg_sq, n, 1.0f - beta1, beta2, 1.0f - beta2, (float)((double)lr / bc1), (float)sqrt(bc2), eps};
    launch_begin();
    if (bits & 15u) {
        uint64_t sb = (n + 255) / 256;
        if (sb > 256 * 32) sb = 256 * 32;
        hipLaunchKernelGGL(k_adam_table_zero_grad_scalar, dim3((uint32_t)sb), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        const uint64_t lb = (n / 4 + 255) / 256 ? (n / 4 + 255) / 256 : 1;
        if (lb > 0x7FFFFFFFull) return NSA_EBADARG;
        if (n >= (1ull << 26)) hipLaunchKernelGGL(k_adam_table_zero_grad<true>, dim3((uint32_t)lb), dim3(256), 0, (hipStream_t)stream, a);
        else                   hipLaunchKernelGGL(k_adam_table_zero_grad<false>, dim3((uint32_t)lb), dim3(256), 0, (hipStream_t)stream, a);
    }
    return launch_end();
}

int nsa_adam_multi_step(const nsa_adam_seg_t* segs, uint32_t count, float beta1, float beta2, float eps, nsa_stream_t stream) {
    using namespace nsa;
    if (count == 0) return NSA_OK;
    if (!segs || count > (uint32_t)ADAM_MULTI_MAX) return NSA_EBADARG;
    AdamMultiArgs a;
    uint32_t blocks = 0;
    for (uint32_t s = 0; s < count; ++s) {
        const nsa_adam_seg_t& g = segs[s];
        if (!g.param || !g.grad || !g.exp_avg || !g.exp_avg_sq || g.step == 0 || g.n == 0 || g.n > (1u << 24)) return NSA_EBADARG;
        if ((reinterpret_cast<uintptr_t>(g.param) | reinterpret_cast<uintptr_t>(g.grad) | reinterpret_cast<uintptr_t>(g.exp_avg) |
             reinterpret_cast<uintptr_t>(g.exp_avg_sq)) & 3u) return NSA_EBADARG;
        a.p[s] = g.param; a.g[s] = const_cast<float*>(g.grad); a.m[s] = g.exp_avg; a.v[s] = g.exp_avg_sq;
        a.n[s] = g.n;
        a.block0[s] = blocks;
        blocks += (g.n + 255) / 256;
        const double bc1 = 1.0 - pow((double)beta1, (double)g.step);
        const double bc2 = 1.0 - pow((double)beta2, (double)g.step);
        a.step_size[s] = (float)((double)g.lr / bc1);
        a.bc2_sqrt[s] = (float)sqrt(bc2);
    }
    a.block0[count] = blocks;
    a.w1 = 1.0f - beta1; a.beta2 = beta2; a.w2 = 1.0f - beta2; a.eps = eps;
    a.count = count;
    launch_begin();
    hipLaunchKernelGGL(k_adam_multi, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return launch_end();
}

int nsa_fill_zero(float* p, uint64_t n, nsa_stream_t stream) {
    using namespace nsa;
    if (!p || (reinterpret_cast<uintptr_t>(p) & 15u)) return NSA_EBADARG;
    if (n == 0) return NSA_OK;
    uint64_t blocks = (n / 4 + 255) / 256;
    if (blocks == 0) blocks = 1;
    if (blocks > 0x7FFFFFFFull) return NSA_EBADARG;
    launch_begin();
    hipLaunchKernelGGL(k_fill_zero, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, p, n);
    return launch_end();
}

static int wn_args(const nsa_wn_layer_t* layers, uint32_t n_layers, nsa::WnArgs& a) {
    using namespace nsa;
    if (!layers || n_layers < 1 || n_layers > (uint32_t)WN_MAX_LAYERS) return NSA_EBADARG;
    uint64_t row = 0, off = 0;
    for (uint32_t l = 0; l < n_layers; ++l) {
        const nsa_wn_layer_t& L = layers[l];
        if (!L.weight_v || !L.weight_g || !L.bias || L.rows == 0 || L.cols == 0) return NSA_EBADARG;
        a.v[l] = L.weight_v; a.g[l] = L.weight_g; a.bias[l] = L.bias;
        a.rows[l] = L.rows; a.cols[l] = L.cols;
        a.row0[l] = (uint32_t)row; a.off[l] = (uint32_t)off;
        row += L.rows;
        off += (uint64_t)L.rows * L.cols + L.rows;
        if (off > 0x7FFFFFFFull) return NSA_EBADARG;
    }
    a.row0[n_layers] = (uint32_t)row; a.off[n_layers] = (uint32_t)off;
    a.n = n_layers;
    a.flat = nullptr; a.norms = nullptr; a.g_flat = nullptr; a.g_params = nullptr;
    return NSA_OK;
}

int nsa_weight_norm_flat(const nsa_wn_layer_t* layers, uint32_t n_layers, float* flat, float* norms, nsa_stream_t stream) {
    using namespace nsa;
    WnArgs a;
    const int rc = wn_args(layers, n_layers, a);
    if (rc != NSA_OK) return rc;
    if (!flat || !norms) return NSA_EBADARG;
    a.flat = flat; a.norms = norms;
    launch_begin();
    hipLaunchKernelGGL(k_weight_norm_flat, dim3(a.row0[a.n] + 1), dim3(64), 0, (hipStream_t)stream, a);
    return launch_end();
}

int nsa_weight_norm_flat_backward(const nsa_wn_layer_t* layers, uint32_t n_layers, const float* norms, const float* g_flat,
                                  float* g_params, nsa_stream_t stream) {
    using namespace nsa;
    WnArgs a;
    const int rc = wn_args(layers, n_layers, a);
    if (rc != NSA_OK) return rc;
    if (!norms || !g_flat || !g_params) return NSA_EBADARG;
    a.norms = const_cast<float*>(norms); a.g_flat = g_flat; a.g_params = g_params;
    launch_begin();
    hipLaunchKernelGGL(k_weight_norm_flat_bwd, dim3(a.row0[a.n]), dim3(64), 0, (hipStream_t)stream, a);
    return launch_end();
}

int nsa_emit_row(float* dst, const float* src, const int32_t* order, uint32_t P, uint64_t n, float fill, nsa_stream_t stream) {
    using namespace nsa;
    if (!dst || P > n) return NSA_EBADARG;
    if (n == 0) return NSA_OK;
    launch_begin();
    hipLaunchKernelGGL(k_emit_row, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, src, order, P, n, fill);
    return launch_end();
}

}  // extern "C"
