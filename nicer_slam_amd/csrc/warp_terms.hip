// warp_terms.hip -- the keyframe re-projection blocks of a mapping iteration (C ABI section 5):
//   * patch warp   (reference code/model/network.py:167-279, uv2patch code/utils/general.py:129-145): the p x p patch around
//                  every sampled pixel of every keyframe is lifted with the ray's RENDERED depth, projected into every keyframe,
//                  and that keyframe's full image is sampled there (F.grid_sample, bilinear, zeros, align_corners=True); the
//                  same patch is read straight from its own image; masks: inside both images, in front of the target camera,
//                  and (p > 1) ground-truth depth of the patch locally flat (variance < 0.01);
//   * flow         (network.py:153-165): the rendered 3-D point of frame idii[e] projected into frame idjj[e], minus its pixel;
//   * masked L1    (code/model/loss.py:136-142 `(a[mask] - b[mask]).abs().mean()`, :106-111 flow L1 on flow_mask).
// The reference builds these from ~40 torch launches per patch size (plus a python loop over the keyframes with boolean-mask
// indexing, i.e. host synchronisations); here: one forward launch, one or two backward launches per block.
//
// Geometry shared by all kernels (rend_util.py:68-93,107-129): pixel (u,v) -> camera ray c = lift(K,u,v,1); world ray
// w = (P [c,1]) - o, d = w / |w|^2 (NOT unit length); x = o + depth d; camera point q = W_t [x,1] (W_t = inverse pose of the
// target, formed by the caller with torch.linalg.inv like the reference); pr = K_t[:3,:3] q; (tu,tv) = pr.xy / (pr.z + 1e-8).
//
// Backward: to the rendered depth [b,n] always; with want_pose also to the source pose (through o and d) and to W_t.  A thread
// owns one (source image, pixel, patch cell) and loops over the targets, so every depth gradient is a fixed-order sum; the pose
// gradients are wave partials added in a fixed order by k_pose_finish (deterministic, no atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/nicer_slam_amd.h"
#include "grid_common.hpp"

namespace nsa {

constexpr int WT = 256;          // threads per block

struct WarpArgs {
    nsa_warp_t in;
    uint32_t patch, p2;
    // forward outputs
    float* sampled;              // [b_t, b_s, n, p2, 3]
    uint8_t* mask;               // [b_t, b_s, n, p2]
    float* gt_rgb;               // [b_t, b_s, n, p2, 3] (the source patch, replicated over targets)
    uint8_t* flat;               // [b_s, n] (patch > 1)
    // backward
    const float* g_sampled;      // [b_t, b_s, n, p2, 3]
    float* g_depth;              // [b_s, n]
    float* g_cell;               // [b_s, n, p2] per-cell partials (patch > 1)
    float* part_src;             // [b_s][waves_per_image][12]
    float* part_w2c;             // [b_s * waves_per_image][slots][12]
    uint32_t waves_per_image;
    int want_pose;
    // flow
    const int64_t* idii;
    const int64_t* idjj;
    uint32_t ne;
    float* flow;                 // [ne, n, 2]
    const float* g_flow;
};

__device__ __forceinline__ void lift_px(const float* __restrict__ K, float u, float v, float (&c)[3]) {
    const float fx = K[0], sk = K[1], cx = K[2], fy = K[5], cy = K[6];
    c[0] = (u - cx + cy * sk / fy - sk * v / fy) / fx;       // rend_util.py:117-125 (z = 1)
    c[1] = (v - cy) / fy;
    c[2] = 1.0f;
}

struct SrcRay {
    float c[3];      // camera-frame pixel ray
    float w[3];      // world-frame ray before the normalisation
    float s;         // |w|^2
    float d[3];      // w / s
    float o[3];
    float x[3];      // o + depth d
};

__device__ __forceinline__ void source_point(const float* __restrict__ P, const float* __restrict__ K, float u, float v, float depth,
                                             SrcRay& r) {
    lift_px(K, u, v, r.c);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float t = P[4 * k] * r.c[0] + P[4 * k + 1] * r.c[1] + P[4 * k + 2] * r.c[2] + P[4 * k + 3];
        r.o[k] = P[4 * k + 3];
        r.w[k] = t - r.o[k];
    }
    r.s = r.w[0] * r.w[0] + r.w[1] * r.w[1] + r.w[2] * r.w[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r.d[k] = r.w[k] / r.s;
        r.x[k] = r.o[k] + depth * r.d[k];
    }
}

struct Proj {
    float q[3];      // camera point in the target
    float tz, tu, tv;
};

__device__ __forceinline__ void project(const float* __restrict__ Wt, const float* __restrict__ Kt, const float (&x)[3], Proj& p) {
    float pr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) p.q[k] = Wt[4 * k] * x[0] + Wt[4 * k + 1] * x[1] + Wt[4 * k + 2] * x[2] + Wt[4 * k + 3];
#pragma unroll
    for (int k = 0; k < 3; ++k) pr[k] = Kt[4 * k] * p.q[0] + Kt[4 * k + 1] * p.q[1] + Kt[4 * k + 2] * p.q[2];
    p.tz = pr[2];
    const float den = pr[2] + 1e-8f;
    p.tu = pr[0] / den;
    p.tv = pr[1] / den;
}

// d(tu,tv) -> d x (returned) and d W_t (12 values, accumulated into gW when non-null)
__device__ __forceinline__ void project_backward(const float* __restrict__ Wt, const float* __restrict__ Kt, const float (&x)[3],
                                                 const Proj& p, float g_tu, float g_tv, float (&gx)[3], float* gW) {
    const float den = p.tz + 1e-8f;
    float gpr[3] = {g_tu / den, g_tv / den, -(g_tu * p.tu + g_tv * p.tv) / den};
    float gq[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) gq[j] = Kt[j] * gpr[0] + Kt[4 + j] * gpr[1] + Kt[8 + j] * gpr[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) gx[j] = Wt[j] * gq[0] + Wt[4 + j] * gq[1] + Wt[8 + j] * gq[2];
    if (gW) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int j = 0; j < 3; ++j) gW[4 * k + j] += gq[k] * x[j];
            gW[4 * k + 3] += gq[k];
        }
    }
}

// d x -> d depth (returned) and d source pose (12 values accumulated into gP when non-null)
__device__ __forceinline__ float source_backward(const SrcRay& r, float depth, const float (&gx)[3], float* gP) {
    const float gdep = gx[0] * r.d[0] + gx[1] * r.d[1] + gx[2] * r.d[2];
    if (gP) {
        float gd[3] = {depth * gx[0], depth * gx[1], depth * gx[2]};
        const float wg = r.w[0] * gd[0] + r.w[1] * gd[1] + r.w[2] * gd[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float gw = gd[k] / r.s - 2.0f * r.w[k] * wg / (r.s * r.s);        // d = w / (w.w)
#pragma unroll
            for (int j = 0; j < 3; ++j) gP[4 * k + j] += gw * r.c[j];
            gP[4 * k + 3] += gx[k];                                                 // o = pose[:3,3]; (P[c,1] - o) cancels it in w
        }
    }
    return gdep;
}

__device__ __forceinline__ const float* frame_image(const nsa_warp_t& L, const float* base, uint32_t bi, uint32_t channels) {
    const uint32_t f = L.frame_index ? (uint32_t)L.frame_index[bi] : bi;
    return base + (size_t)f * L.H * L.W * channels;
}

struct Bilinear {
    int x0, y0;
    float fx1, fy1, fx0, fy0;     // ix_se - ix, iy_se - iy, ix - ix_nw, iy - iy_nw
    bool ok;
};

// F.grid_sample(align_corners=True) pixel coordinates of the normalised grid the reference forms (network.py:213-217)
__device__ __forceinline__ Bilinear bilinear_setup(const nsa_warp_t& L, float gx, float gy) {
    Bilinear b;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(L.W - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(L.H - 1);
    b.ok = fabsf(ix) < 1.0e8f && fabsf(iy) < 1.0e8f;          // (non-finite / absurd coordinates sample nothing)
    const float fx = b.ok ? floorf(ix) : 0.0f, fy = b.ok ? floorf(iy) : 0.0f;
    b.x0 = (int)fx;
    b.y0 = (int)fy;
    b.fx1 = (fx + 1.0f) - ix;
    b.fy1 = (fy + 1.0f) - iy;
    b.fx0 = ix - fx;
    b.fy0 = iy - fy;
    return b;
}

__device__ __forceinline__ void texel(const nsa_warp_t& L, const float* img, int x, int y, bool ok, float (&v)[3]) {
    if (ok && x >= 0 && y >= 0 && x < (int)L.W && y < (int)L.H) {
        const float* p = img + ((size_t)y * L.W + x) * 3;
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
    } else {
        v[0] = v[1] = v[2] = 0.0f;
    }
}

__device__ __forceinline__ void patch_cell(uint32_t patch, uint32_t c, float& du, float& dv) {
    // general.py:139-144: meshgrid(x, y, indexing="ij") -> cell (ix, iy) = (c / p, c % p), offset (ix - half, iy - half) on (u, v)
    const int half = (int)patch / 2;
    du = patch > 1 ? (float)((int)(c / patch) - half) : 0.0f;
    dv = patch > 1 ? (float)((int)(c % patch) - half) : 0.0f;
}

// ---- patch warp: forward.  One thread per (target, source, pixel, cell). -----------------------------------------
__global__ __launch_bounds__(WT) void k_warp_fwd(WarpArgs a) {
    const nsa_warp_t& L = a.in;
    const uint64_t total = (uint64_t)L.b * L.b * L.n * a.p2;
    const uint64_t idx = (uint64_t)blockIdx.x * WT + threadIdx.x;
    if (idx >= total) return;
    const uint32_t c = (uint32_t)(idx % a.p2);
    const uint32_t i = (uint32_t)((idx / a.p2) % L.n);
    const uint32_t s = (uint32_t)((idx / ((uint64_t)a.p2 * L.n)) % L.b);
    const uint32_t t = (uint32_t)(idx / ((uint64_t)a.p2 * L.n * L.b));
    float du, dv;
    patch_cell(a.patch, c, du, dv);
    const float u = L.uv[2 * ((size_t)s * L.n + i)] + du, v = L.uv[2 * ((size_t)s * L.n + i) + 1] + dv;
    SrcRay r;
    source_point(L.pose + 16 * s, L.K + 16 * s, u, v, L.depth[(size_t)s * L.n + i], r);
    Proj p;
    project(L.w2c + 16 * t, L.K + 16 * t, r.x, p);
    const float gx = p.tu / (float)L.W * 2.0f - 1.0f, gy = p.tv / (float)L.H * 2.0f - 1.0f;     // network.py:213-215
    const Bilinear bl = bilinear_setup(L, gx, gy);
    const float* img = frame_image(L, L.images, t, 3);
    float nw[3], ne[3], sw[3], se[3];
    texel(L, img, bl.x0, bl.y0, bl.ok, nw);
    texel(L, img, bl.x0 + 1, bl.y0, bl.ok, ne);
    texel(L, img, bl.x0, bl.y0 + 1, bl.ok, sw);
    texel(L, img, bl.x0 + 1, bl.y0 + 1, bl.ok, se);
    const float wnw = bl.fx1 * bl.fy1, wne = bl.fx0 * bl.fy1, wsw = bl.fx1 * bl.fy0, wse = bl.fx0 * bl.fy0;
#pragma unroll
    for (int k = 0; k < 3; ++k) a.sampled[3 * idx + k] = nw[k] * wnw + ne[k] * wne + sw[k] * wsw + se[k] * wse;
    const bool tmask = gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f && p.tz > 0.0f;       // network.py:224-230
    // the patch itself, read from its own image; ones outside (network.py:240-253)
    const bool inside = 0.0f <= u && 0.0f <= v && u < (float)L.W && v < (float)L.H;
    float g[3] = {1.0f, 1.0f, 1.0f};
    if (inside) {
        const float* src = frame_image(L, L.images, s, 3) + ((size_t)(int64_t)v * L.W + (size_t)(int64_t)u) * 3;   // .long(): truncation
        g[0] = src[0]; g[1] = src[1]; g[2] = src[2];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) a.gt_rgb[3 * idx + k] = g[k];
    const bool flat = a.patch > 1 ? a.flat[(size_t)s * L.n + i] != 0 : true;
    a.mask[idx] = (tmask && inside && flat) ? 1 : 0;
}

// ---- flat-depth test of the source patches (network.py:259-270): biased variance of the p^2 ground-truth depths < 0.01
__global__ __launch_bounds__(WT) void k_warp_flat(WarpArgs a) {
    const nsa_warp_t& L = a.in;
    const uint32_t r = blockIdx.x * WT + threadIdx.x;
    if (r >= L.b * L.n) return;
    const uint32_t s = r / L.n;
    const float* dep = frame_image(L, L.depths, s, 1);
    const float u0 = L.uv[2 * (size_t)r], v0 = L.uv[2 * (size_t)r + 1];
    float sum = 0.0f;
    for (uint32_t c = 0; c < a.p2; ++c) {
        float du, dv;
        patch_cell(a.patch, c, du, dv);
        const float u = u0 + du, v = v0 + dv;
        const bool inside = 0.0f <= u && 0.0f <= v && u < (float)L.W && v < (float)L.H;
        sum += inside ? dep[(size_t)(int64_t)v * L.W + (size_t)(int64_t)u] : 1.0f;
    }
    const float mean = sum / (float)a.p2;
    float ss = 0.0f;
    for (uint32_t c = 0; c < a.p2; ++c) {
        float du, dv;
        patch_cell(a.patch, c, du, dv);
        const float u = u0 + du, v = v0 + dv;
        const bool inside = 0.0f <= u && 0.0f <= v && u < (float)L.W && v < (float)L.H;
        const float x = (inside ? dep[(size_t)(int64_t)v * L.W + (size_t)(int64_t)u] : 1.0f) - mean;
        ss += x * x;
    }
    a.flat[r] = (ss / (float)a.p2 < 0.01f) ? 1 : 0;
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

// ---- patch warp: backward.  grid (ceil(n p2 / WT), b_s); a thread owns (s, pixel, cell) and loops over the targets. ----
__global__ __launch_bounds__(WT) void k_warp_bwd(WarpArgs a) {
    const nsa_warp_t& L = a.in;
    const uint32_t s = blockIdx.y;
    const uint32_t j0 = blockIdx.x * WT + threadIdx.x;
    const uint32_t per = L.n * a.p2;
    const bool live = j0 < per;
    const uint32_t j = live ? j0 : per - 1;
    const uint32_t i = j / a.p2, c = j % a.p2;
    float du, dv;
    patch_cell(a.patch, c, du, dv);
    const float u = L.uv[2 * ((size_t)s * L.n + i)] + du, v = L.uv[2 * ((size_t)s * L.n + i) + 1] + dv;
    const float depth = L.depth[(size_t)s * L.n + i];
    SrcRay r;
    source_point(L.pose + 16 * s, L.K + 16 * s, u, v, depth, r);
    const uint32_t wave_in_image = blockIdx.x * (WT / 64) + (threadIdx.x >> 6);
    const uint32_t wave_global = s * a.waves_per_image + wave_in_image;
    const uint32_t lane = threadIdx.x & 63;
    float gxs[3] = {0.0f, 0.0f, 0.0f};
    for (uint32_t t = 0; t < L.b; ++t) {
        Proj p;
        project(L.w2c + 16 * t, L.K + 16 * t, r.x, p);
        const float gx = p.tu / (float)L.W * 2.0f - 1.0f, gy = p.tv / (float)L.H * 2.0f - 1.0f;
        const Bilinear bl = bilinear_setup(L, gx, gy);
        const float* img = frame_image(L, L.images, t, 3);
        float nw[3], ne[3], sw[3], se[3];
        texel(L, img, bl.x0, bl.y0, bl.ok, nw);
        texel(L, img, bl.x0 + 1, bl.y0, bl.ok, ne);
        texel(L, img, bl.x0, bl.y0 + 1, bl.ok, sw);
        texel(L, img, bl.x0 + 1, bl.y0 + 1, bl.ok, se);
        const float* g = a.g_sampled + 3 * ((((size_t)t * L.b + s) * L.n + i) * a.p2 + c);
        float gix = 0.0f, giy = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float go = live ? g[k] : 0.0f;
            gix += go * (-nw[k] * bl.fy1 + ne[k] * bl.fy1 - sw[k] * bl.fy0 + se[k] * bl.fy0);
            giy += go * (-nw[k] * bl.fx1 - ne[k] * bl.fx0 + sw[k] * bl.fx1 + se[k] * bl.fx0);
        }
        // grid = tuv / (W, H) * 2 - 1 ; pixel = (grid + 1) / 2 * (size - 1)
        const float g_tu = gix * ((float)(L.W - 1) / 2.0f) * (2.0f / (float)L.W);
        const float g_tv = giy * ((float)(L.H - 1) / 2.0f) * (2.0f / (float)L.H);
        float gW[12], gx3[3];
#pragma unroll
        for (int q = 0; q < 12; ++q) gW[q] = 0.0f;
        project_backward(L.w2c + 16 * t, L.K + 16 * t, r.x, p, g_tu, g_tv, gx3, a.want_pose ? gW : nullptr);
#pragma unroll
        for (int k = 0; k < 3; ++k) gxs[k] += gx3[k];
        if (a.want_pose) {
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const float x = wave_sum(gW[q]);
                if (lane == 0) a.part_w2c[((size_t)wave_global * L.b + t) * 12 + q] = x;
            }
        }
    }
    float gP[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) gP[q] = 0.0f;
    const float gdep = source_backward(r, depth, gxs, a.want_pose ? gP : nullptr);
    if (live) {
        if (a.p2 == 1) a.g_depth[(size_t)s * L.n + i] = gdep;
        else a.g_cell[((size_t)s * L.n + i) * a.p2 + c] = gdep;
    }
    if (a.want_pose) {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const float x = wave_sum(gP[q]);
            if (lane == 0) a.part_src[((size_t)s * a.waves_per_image + wave_in_image) * 12 + q] = x;
        }
    }
}

// g_depth[s,i] = sum over the patch cells, in cell order
__global__ __launch_bounds__(WT) void k_warp_cells(WarpArgs a) {
    const nsa_warp_t& L = a.in;
    const uint32_t r = blockIdx.x * WT + threadIdx.x;
    if (r >= L.b * L.n) return;
    float acc = 0.0f;
    for (uint32_t c = 0; c < a.p2; ++c) acc += a.g_cell[(size_t)r * a.p2 + c];
    a.g_depth[r] = acc;
}

// Pose gradients from the wave partials, fixed order.  grid (b, 2): y = 0 -> g_pose[x] from part_src[x][*]; y = 1 -> g_w2c[x]
// from part_w2c[*][slot] for every slot whose target is x (slot_target == NULL: slot == target).
struct PoseFinishArgs {
    const float* part_src;
    const float* part_w2c;
    uint32_t b, waves_per_image, slots;
    const int64_t* slot_target;
    float* g_pose;       // [b,4,4]
    float* g_w2c;        // [b,4,4]
};

__global__ __launch_bounds__(WT) void k_pose_finish(PoseFinishArgs a) {
    __shared__ float red[21][12];
    const uint32_t x = blockIdx.x;
    const uint32_t grp = threadIdx.x / 12, q = threadIdx.x % 12;
    float acc = 0.0f;
    if (grp < 21) {
        if (blockIdx.y == 0) {
            for (uint32_t w = grp; w < a.waves_per_image; w += 21) acc += a.part_src[((size_t)x * a.waves_per_image + w) * 12 + q];
        } else {
            const uint32_t waves = a.b * a.waves_per_image;
            for (uint32_t sl = 0; sl < a.slots; ++sl) {
                const uint32_t tgt = a.slot_target ? (uint32_t)a.slot_target[sl] : sl;
                if (tgt != x) continue;
                for (uint32_t w = grp; w < waves; w += 21) acc += a.part_w2c[((size_t)w * a.slots + sl) * 12 + q];
            }
        }
        red[grp][q] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        float v = 0.0f;
        if (threadIdx.x < 12)
            for (int g = 0; g < 21; ++g) v += red[g][threadIdx.x];
        (blockIdx.y == 0 ? a.g_pose : a.g_w2c)[16 * x + threadIdx.x] = v;        // bottom row: zero
    }
}

// ---- flow: forward, one thread per (edge, pixel) --------------------------------------------------------------------
__global__ __launch_bounds__(WT) void k_flow_fwd(WarpArgs a) {
    const nsa_warp_t& L = a.in;
    const uint64_t idx = (uint64_t)blockIdx.x * WT + threadIdx.x;
    if (idx >= (uint64_t)a.ne * L.n) return;
    const uint32_t e = (uint32_t)(idx / L.n), i = (uint32_t)(idx % L.n);
    const uint32_t s = (uint32_t)a.idii[e], t = (uint32_t)a.idjj[e];
    const float u = L.uv[2 * ((size_t)s * L.n + i)], v = L.uv[2 * ((size_t)s * L.n + i) + 1];
    SrcRay r;
    source_point(L.pose + 16 * s, L.K + 16 * s, u, v, L.depth[(size_t)s * L.n + i], r);
    Proj p;
    project(L.w2c + 16 * t, L.K + 16 * t, r.x, p);
    a.flow[2 * idx] = p.tu - u;
    a.flow[2 * idx + 1] = p.tv - v;
}

// ---- flow: backward, grid (ceil(n / WT), b); a thread owns (s, pixel) and loops over the edges leaving s -------------
__global__ __launch_bounds__(WT) void k_flow_bwd(WarpArgs a) {
    const nsa_warp_t& L = a.in;
    const uint32_t s = blockIdx.y;
    const uint32_t i0 = blockIdx.x * WT + threadIdx.x;
    const bool live = i0 < L.n;
    const uint32_t i = live ? i0 : L.n - 1;
    const float u = L.uv[2 * ((size_t)s * L.n + i)], v = L.uv[2 * ((size_t)s * L.n + i) + 1];
    const float depth = L.depth[(size_t)s * L.n + i];
    SrcRay r;
    source_point(L.pose + 16 * s, L.K + 16 * s, u, v, depth, r);
    const uint32_t wave_in_image = blockIdx.x * (WT / 64) + (threadIdx.x >> 6);
    const uint32_t wave_global = s * a.waves_per_image + wave_in_image;
    const uint32_t lane = threadIdx.x & 63;
    float gxs[3] = {0.0f, 0.0f, 0.0f};
    for (uint32_t e = 0; e < a.ne; ++e) {
        const bool mine = (uint32_t)a.idii[e] == s;                     // uniform over the block
        float gW[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) gW[q] = 0.0f;
        if (mine) {
            const uint32_t t = (uint32_t)a.idjj[e];
            Proj p;
            project(L.w2c + 16 * t, L.K + 16 * t, r.x, p);
            const float* g = a.g_flow + 2 * ((size_t)e * L.n + i);
            float gx3[3];
            project_backward(L.w2c + 16 * t, L.K + 16 * t, r.x, p, live ? g[0] : 0.0f, live ? g[1] : 0.0f, gx3,
                             a.want_pose ? gW : nullptr);
#pragma unroll
            for (int k = 0; k < 3; ++k) gxs[k] += gx3[k];
        }
        if (a.want_pose) {
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const float x = wave_sum(gW[q]);
                if (lane == 0) a.part_w2c[((size_t)wave_global * a.ne + e) * 12 + q] = x;
            }
        }
    }
    float gP[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) gP[q] = 0.0f;
    const float gdep = source_backward(r, depth, gxs, a.want_pose ? gP : nullptr);
    if (live) a.g_depth[(size_t)s * L.n + i] = gdep;
    if (a.want_pose) {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const float x = wave_sum(gP[q]);
            if (lane == 0) a.part_src[((size_t)s * a.waves_per_image + wave_in_image) * 12 + q] = x;
        }
    }
}

// ---- masked L1 mean --------------------------------------------------------------------------------------------------
struct L1Args {
    const float* pred;
    const float* target;
    const uint8_t* mask;      // [items] or NULL
    uint64_t items;
    uint32_t ch, blocks;
    double* part;             // [blocks][2] ; part[2*blocks .. +1] = (sum, count) totals
    float* loss;              // [1]
    float* g_pred;            // [items, ch] or NULL
};

__global__ __launch_bounds__(WT) void k_l1_part(L1Args a) {
    __shared__ double red[2][4];
    double s = 0.0, cnt = 0.0;
    for (uint64_t it = (uint64_t)blockIdx.x * WT + threadIdx.x; it < a.items; it += (uint64_t)a.blocks * WT) {
        if (a.mask && !a.mask[it]) continue;
        cnt += 1.0;
        for (uint32_t k = 0; k < a.ch; ++k) s += (double)fabsf(a.pred[it * a.ch + k] - a.target[it * a.ch + k]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off); cnt += __shfl_down(cnt, off); }
    if (lane == 0) { red[0][wave] = s; red[1][wave] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.part[2 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        a.part[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

__global__ __launch_bounds__(64) void k_l1_final(L1Args a) {          // one wave, fixed order (deterministic)
    double s = 0.0, cnt = 0.0;
    for (uint32_t k = threadIdx.x; k < a.blocks; k += 64) { s += a.part[2 * k]; cnt += a.part[2 * k + 1]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); cnt += __shfl_xor(cnt, off); }
    if (threadIdx.x != 0) return;
    a.part[2 * a.blocks] = s;
    a.part[2 * a.blocks + 1] = cnt;
    a.loss[0] = (float)(s / (cnt * (double)a.ch));         // an empty selection gives NaN, like torch's mean of nothing
}

__global__ __launch_bounds__(WT) void k_l1_grad(L1Args a) {
    const uint64_t e = (uint64_t)blockIdx.x * WT + threadIdx.x;
    if (e >= a.items * a.ch) return;
    const uint64_t it = e / a.ch;
    float g = 0.0f;
    if (!a.mask || a.mask[it]) {
        const float d = a.pred[e] - a.target[e];
        const float inv = 1.0f / (float)(a.part[2 * a.blocks + 1] * (double)a.ch);
        g = d > 0.0f ? inv : (d < 0.0f ? -inv : 0.0f);
    }
    a.g_pred[e] = g;
}

static inline uint32_t waves_per_image(uint32_t per_image) { return ((per_image + WT - 1) / WT) * (WT / 64); }

static inline bool warp_ok(const nsa_warp_t* in) {
    return in && in->b && in->n && in->uv && in->pose && in->w2c && in->K && in->depth;     // (H, W: patch warp only)
}

}  // namespace nsa

extern "C" {

int nsa_patch_warp_forward(const nsa_warp_t* in, uint32_t patch, float* sampled, uint8_t* mask, float* gt_rgb, uint8_t* flat,
                           nsa_stream_t stream) {
    using namespace nsa;
    if (!warp_ok(in) || in->H < 2 || in->W < 2 || !in->images || !patch || !(patch & 1) || !sampled || !mask || !gt_rgb) return NSA_EBADARG;
    if (patch > 1 && (!in->depths || !flat)) return NSA_EBADARG;
    WarpArgs a{};
    a.in = *in;
    a.patch = patch;
    a.p2 = patch * patch;
    a.sampled = sampled;
    a.mask = mask;
    a.gt_rgb = gt_rgb;
    a.flat = flat;
    launch_begin();
    if (patch > 1)
        hipLaunchKernelGGL(k_warp_flat, dim3((in->b * in->n + WT - 1) / WT), dim3(WT), 0, (hipStream_t)stream, a);
    const uint64_t total = (uint64_t)in->b * in->b * in->n * a.p2;
    hipLaunchKernelGGL(k_warp_fwd, dim3((uint32_t)((total + WT - 1) / WT)), dim3(WT), 0, (hipStream_t)stream, a);
    return launch_end();
}

uint64_t nsa_patch_warp_workspace(uint32_t b, uint32_t n, uint32_t patch, int want_pose) {
    using namespace nsa;
    const uint64_t p2 = (uint64_t)patch * patch;
    uint64_t f = patch > 1 ? (uint64_t)b * n * p2 : 0;
    if (want_pose) {
        const uint64_t w = waves_per_image((uint32_t)(n * p2));
        f += (uint64_t)b * w * 12 + (uint64_t)b * w * b * 12;
    }
    return f;
}

int nsa_patch_warp_backward(const nsa_warp_t* in, uint32_t patch, const float* g_sampled, float* g_depth, float* g_pose,
                            float* g_w2c, float* workspace, nsa_stream_t stream) {
    using namespace nsa;
    if (!warp_ok(in) || in->H < 2 || in->W < 2 || !in->images || !patch || !(patch & 1) || !g_sampled || !g_depth) return NSA_EBADARG;
    const int want_pose = g_pose != nullptr || g_w2c != nullptr;
    if (want_pose && (!g_pose || !g_w2c)) return NSA_EBADARG;
    if ((patch > 1 || want_pose) && !workspace) return NSA_EBADARG;
    WarpArgs a{};
    a.in = *in;
    a.patch = patch;
    a.p2 = patch * patch;
    a.g_sampled = g_sampled;
    a.g_depth = g_depth;
    a.want_pose = want_pose;
    const uint32_t per = in->n * a.p2;
    a.waves_per_image = waves_per_image(per);
    float* ws = workspace;
    if (patch > 1) { a.g_cell = ws; ws += (size_t)in->b * in->n * a.p2; }
    if (want_pose) {
        a.part_src = ws;
        a.part_w2c = ws + (size_t)in->b * a.waves_per_image * 12;
    }
    launch_begin();
    hipLaunchKernelGGL(k_warp_bwd, dim3((per + WT - 1) / WT, in->b), dim3(WT), 0, (hipStream_t)stream, a);
    if (patch > 1)
        hipLaunchKernelGGL(k_warp_cells, dim3((in->b * in->n + WT - 1) / WT), dim3(WT), 0, (hipStream_t)stream, a);
    if (want_pose) {
        PoseFinishArgs f{a.part_src, a.part_w2c, in->b, a.waves_per_image, in->b, nullptr, g_pose, g_w2c};
        hipLaunchKernelGGL(k_pose_finish, dim3(in->b, 2), dim3(WT), 0, (hipStream_t)stream, f);
    }
    return launch_end();
}

int nsa_flow_forward(const nsa_warp_t* in, const int64_t* idii, const int64_t* idjj, uint32_t ne, float* flow,
                     nsa_stream_t stream) {
    using namespace nsa;
    if (!warp_ok(in) || (ne && (!idii || !idjj || !flow))) return NSA_EBADARG;
    if (!ne) return NSA_OK;
    WarpArgs a{};
    a.in = *in;
    a.idii = idii;
    a.idjj = idjj;
    a.ne = ne;
    a.flow = flow;
    launch_begin();
    const uint64_t total = (uint64_t)ne * in->n;
    hipLaunchKernelGGL(k_flow_fwd, dim3((uint32_t)((total + WT - 1) / WT)), dim3(WT), 0, (hipStream_t)stream, a);
    return launch_end();
}

uint64_t nsa_flow_workspace(uint32_t b, uint32_t n, uint32_t ne, int want_pose) {
    using namespace nsa;
    if (!want_pose) return 0;
    const uint64_t w = waves_per_image(n);
    return (uint64_t)b * w * 12 + (uint64_t)b * w * ne * 12;
}

int nsa_flow_backward(const nsa_warp_t* in, const int64_t* idii, const int64_t* idjj, uint32_t ne, const float* g_flow,
                      float* g_depth, float* g_pose, float* g_w2c, float* workspace, nsa_stream_t stream) {
    using namespace nsa;
    if (!warp_ok(in) || !g_depth || (ne && (!idii || !idjj || !g_flow))) return NSA_EBADARG;
    const int want_pose = g_pose != nullptr || g_w2c != nullptr;
    if (want_pose && (!g_pose || !g_w2c || !workspace)) return NSA_EBADARG;
    WarpArgs a{};
    a.in = *in;
    a.idii = idii;
    a.idjj = idjj;
    a.ne = ne;
    a.g_flow = g_flow;
    a.g_depth = g_depth;
    a.want_pose = want_pose;
    a.waves_per_image = waves_per_image(in->n);
    if (want_pose) {
        a.part_src = workspace;
        a.part_w2c = workspace + (size_t)in->b * a.waves_per_image * 12;
    }
    launch_begin();
    hipLaunchKernelGGL(k_flow_bwd, dim3((in->n + WT - 1) / WT, in->b), dim3(WT), 0, (hipStream_t)stream, a);
    if (want_pose) {
        PoseFinishArgs f{a.part_src, a.part_w2c, in->b, a.waves_per_image, ne, idjj, g_pose, g_w2c};
        hipLaunchKernelGGL(k_pose_finish, dim3(in->b, 2), dim3(WT), 0, (hipStream_t)stream, f);
    }
    return launch_end();
}

uint64_t nsa_masked_l1_workspace(uint64_t items) {
    using namespace nsa;
    uint64_t blocks = (items + WT - 1) / WT;
    blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
    return (blocks + 1) * 2 * 2;          // doubles, counted in floats
}

int nsa_masked_l1(const float* pred, const float* target, const uint8_t* mask, uint64_t items, uint32_t channels, float* loss,
                  float* g_pred, float* workspace, nsa_stream_t stream) {
    using namespace nsa;
    if (!loss || !workspace || !channels || (items && (!pred || !target))) return NSA_EBADARG;
    if (reinterpret_cast<uintptr_t>(workspace) & 7u) return NSA_EBADARG;
    L1Args a{};
    a.pred = pred;
    a.target = target;
    a.mask = mask;
    a.items = items;
    a.ch = channels;
    uint64_t blocks = (items + WT - 1) / WT;
    a.blocks = (uint32_t)(blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks));
    a.part = reinterpret_cast<double*>(workspace);
    a.loss = loss;
    a.g_pred = g_pred;
    launch_begin();
    hipLaunchKernelGGL(k_l1_part, dim3(a.blocks), dim3(WT), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_l1_final, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    if (g_pred && items)
        hipLaunchKernelGGL(k_l1_grad, dim3((uint32_t)((items * channels + WT - 1) / WT)), dim3(WT), 0, (hipStream_t)stream, a);
    return launch_end();
}

}  // extern "C"
