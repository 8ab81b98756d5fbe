// loss_terms.hip -- the per-ray terms of the mapping / tracking objective and their gradients in three launches
// (C ABI section 4: nsa_slam_loss).
//
// Reference: SLAMLoss.forward (code/model/loss.py:113-233) with the scale-and-shift-invariant monocular depth loss of
// code/utils/MiDaS.py:6-143 (alpha = 0.5, one scale, batch-based reduction) -- the terms
//     rgb L1 (:57-65,131) | eikonal (:80-84) | smooth (:67-78) | ssi depth (:86-93) | gt-depth L1 (:95-99) | normal L1 + cos (:101-111)
// of which the reference builds ~150 small torch launches plus their autograd graph.  The flow and patch-warp terms stay with
// the host (they gather through boolean masks of data-dependent size).
//
//   k_loss_stats   per image: the five sums of the 2x2 least-squares system for (scale, shift), the mask counts, and the
//                  foreground mask of every ray (its sdf samples change sign, loss.py:164-167); partial sums per block, fp64
//   k_loss_terms   per ray and per eikonal point: every term's contribution and the gradient of the WEIGHTED total with respect
//                  to rgb_values, depth_values, normal_map, grad_theta, grad_theta_nei; block partials of the term sums, fp64
//   k_loss_final   adds the block partials in block order (deterministic) and applies the normalisers
// All reductions are accumulated in fp64 and added in a fixed order; the element-wise arithmetic is fp32 like torch's.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/nicer_slam_amd.h"
#include "grid_common.hpp"

namespace nsa {

constexpr int LT = 256;            // threads per block
constexpr int NSTAT = 8;           // per-image statistics: a00 a01 a11 b0 b1 | (unused) | gt-depth count | (unused)
constexpr int NTERM = 8;           // rgb eik smooth depth_data depth_reg gt_depth nl1 ncos

struct LossArgs {
    nsa_loss_t in;
    double* stat_part;             // [blocks_per_image * bs][NSTAT]
    double* term_part;             // [term blocks][NTERM]
    float* fg;                     // [R] foreground & gt mask (0/1)
    uint32_t blocks_per_image, term_blocks;
};

__device__ __forceinline__ double block_sum(double v, double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];       // same value in every thread, fixed order
}

// grid (blocks_per_image, bs)
__global__ __launch_bounds__(LT) void k_loss_stats(LossArgs a) {
    __shared__ double red[4];
    const nsa_loss_t& L = a.in;
    const uint32_t b = blockIdx.y;
    double s[NSTAT] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * LT + threadIdx.x; i < L.n; i += a.blocks_per_image * LT) {
        const uint32_t r = b * L.n + i;
        bool pos = false, neg = false;
        const float* row = L.sdf + (size_t)r * L.S;
        for (uint32_t k = 0; k < L.S; ++k) {
            const float v = row[k];
            pos = pos || v > 0.0f;
            neg = neg || v < 0.0f;
        }
        const bool m = (L.mask_gt[r] > 0.5f) && pos && neg;
        a.fg[r] = m ? 1.0f : 0.0f;
        const float md = (L.depth_whole_image || m) ? 1.0f : 0.0f;                 // mask of the depth term
        const float p = L.depth[r], t = L.depth_mono[r] * 50.0f + 0.5f;            // loss.py:92
        s[0] += (double)(md * p * p);
        s[1] += (double)(md * p);
        s[2] += (double)md;
        s[3] += (double)(md * p * t);
        s[4] += (double)(md * t);
        s[6] += L.depth_real_mask[r] > 0.0f ? 1.0 : 0.0;
    }
    double* out = a.stat_part + ((size_t)b * a.blocks_per_image + blockIdx.x) * NSTAT;
#pragma unroll
    for (int k = 0; k < NSTAT; ++k) {
        const double v = block_sum(s[k], red);
        if (threadIdx.x == 0) out[k] = v;
    }
}

struct ImageFit {
    float scale, shift;
    double M;
};

// (scale, shift) of image b from the block partials (every thread recomputes the tiny fixed-order sum)
__device__ __forceinline__ ImageFit fit_image(const LossArgs& a, uint32_t b) {
    double s[5] = {0, 0, 0, 0, 0};
    for (uint32_t k = 0; k < a.blocks_per_image; ++k)
#pragma unroll
        for (int j = 0; j < 5; ++j) s[j] += a.stat_part[((size_t)b * a.blocks_per_image + k) * NSTAT + j];
    // MiDaS.py:6-26 in fp32, from fp32 sums like torch's
    const float a00 = (float)s[0], a01 = (float)s[1], a11 = (float)s[2], b0 = (float)s[3], b1 = (float)s[4];
    const float det = a00 * a11 - a01 * a01;
    ImageFit f;
    f.scale = det != 0.0f ? (a11 * b0 - a01 * b1) / det : 0.0f;
    f.shift = det != 0.0f ? (-a01 * b0 + a00 * b1) / det : 0.0f;
    f.M = s[2];
    return f;
}

__device__ __forceinline__ float sgn(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

// residual of the aligned depth at ray r (0 outside the depth mask): d = mask * (scale * p + shift - target)
__device__ __forceinline__ float depth_resid(const nsa_loss_t& L, const float* fg, const ImageFit& f, uint32_t r, float& md) {
    md = (L.depth_whole_image || fg[r] > 0.5f) ? 1.0f : 0.0f;
    const float res = f.scale * L.depth[r] + f.shift - (L.depth_mono[r] * 50.0f + 0.5f);
    return md * res;
}

// one thread per ray (first R threads) and per eikonal point (first E threads)
__global__ __launch_bounds__(LT) void k_loss_terms(LossArgs a) {
    __shared__ double red[4];
    const nsa_loss_t& L = a.in;
    const uint32_t R = L.bs * L.n;
    const uint32_t t = blockIdx.x * LT + threadIdx.x;
    double term[NTERM] = {0, 0, 0, 0, 0, 0, 0, 0};
    // global normalisers (fixed-order sums over the images)
    double M_all = 0.0, cnt_gt = 0.0;
    for (uint32_t b = 0; b < L.bs; ++b)
        for (uint32_t k = 0; k < a.blocks_per_image; ++k) {
            M_all += a.stat_part[((size_t)b * a.blocks_per_image + k) * NSTAT + 2];
            cnt_gt += a.stat_part[((size_t)b * a.blocks_per_image + k) * NSTAT + 6];
        }
    if (t < R) {
        const uint32_t b = t / L.n, i = t % L.n;
        const float m = a.fg[t];
        // ---- rgb: mean |pred - gt| over 3R values
        {
            const float inv = 1.0f / (3.0f * (float)R);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float d = L.rgb[3 * t + c] - L.rgb_gt[3 * t + c];
                term[0] += (double)fabsf(d);
                L.g_rgb[3 * t + c] = L.w_rgb * sgn(d) * inv;
            }
        }
        // ---- depth: ssi data term + alpha * first-difference regulariser along the ray index of the image (MiDaS.py:29-143)
        float g_depth = 0.0f;
        if (L.w_depth > 0.0f) {
            const ImageFit f = fit_image(a, b);
            float md, ml, mr;
            const float d = depth_resid(L, a.fg, f, t, md);
            term[3] += (double)(d * d);                                   // md * res^2 (md is 0/1)
            const float data_div = (float)(2.0 * M_all), reg_div = (float)M_all;
            float gres = data_div != 0.0f ? 2.0f * d / data_div : 0.0f;   // d/d res of sum(md res^2) / (2 sum M)
            float gd = 0.0f;                                              // d/d d_i of the regulariser sum
            if (i + 1 < L.n) {
                const float dr = depth_resid(L, a.fg, f, t + 1, mr);
                const float w = md * mr;
                term[4] += (double)(fabsf(dr - d) * w);                   // pair (i, i+1), counted once
                gd -= sgn(dr - d) * w;
            }
            if (i > 0) {
                const float dl = depth_resid(L, a.fg, f, t - 1, ml);
                gd += sgn(d - dl) * (ml * md);
            }
            if (reg_div != 0.0f) gres += 0.5f * md * gd / reg_div;
            g_depth = L.w_depth * gres * f.scale;
        }
        // ---- gt depth: masked mean |pred - target| (mean over an empty selection is NaN, like torch)
        if (L.w_gtdepth > 0.0f) {
            if (L.depth_real_mask[t] > 0.0f) {
                const float d = L.depth[t] - L.depth_real[t];
                term[5] += (double)fabsf(d);
                g_depth += L.w_gtdepth * sgn(d) / (float)cnt_gt;
            }
        }
        L.g_depth[t] = g_depth;
        // ---- normals: p = normalize(m * n_pred), g = normalize(m * n_gt)  (F.normalize: v / max(|v|, 1e-12))
        {
            float v[3], gt[3], p[3], g[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { v[c] = L.normal[3 * t + c] * m; gt[c] = L.normal_gt[3 * t + c] * m; }
            const float nv = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
            const float ng = fmaxf(sqrtf(gt[0] * gt[0] + gt[1] * gt[1] + gt[2] * gt[2]), 1e-12f);
            float dot = 0.0f, l1 = 0.0f, u[3];
            const float invR = 1.0f / (float)R;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                p[c] = v[c] / nv; g[c] = gt[c] / ng;
                dot += p[c] * g[c];
                l1 += fabsf(p[c] - g[c]);
                u[c] = (L.w_nl1 * sgn(p[c] - g[c]) - L.w_ncos * g[c]) * invR;      // d total / d p
            }
            term[6] += (double)l1;
            term[7] += (double)(1.0f - dot);
            const float pu = p[0] * u[0] + p[1] * u[1] + p[2] * u[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) L.g_normal[3 * t + c] = m * (u[c] - p[c] * pu) / nv;
        }
    }
    if (t < L.E) {
        float g[3], h[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { g[c] = L.grad_theta[3 * t + c]; h[c] = L.grad_theta_nei ? L.grad_theta_nei[3 * t + c] : 0.0f; }
        const float n = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
        const float invE = 1.0f / (float)L.E;
        float out[3] = {0.0f, 0.0f, 0.0f}, outn[3] = {0.0f, 0.0f, 0.0f};
        if (L.w_eik > 0.0f) {                                   // mean (|g| - 1)^2
            term[1] += (double)((n - 1.0f) * (n - 1.0f));
            const float k = n > 0.0f ? L.w_eik * 2.0f * (n - 1.0f) / n * invE : 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) out[c] += k * g[c];
        }
        if (L.w_smooth > 0.0f && L.grad_theta_nei) {            // mean | g / (|g| + 1e-5) - h / (|h| + 1e-5) |
            const float nh = sqrtf(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
            float d[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) d[c] = g[c] / (n + 1e-5f) - h[c] / (nh + 1e-5f);
            const float nd = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            term[2] += (double)nd;
            if (nd > 0.0f) {
                float q[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) q[c] = L.w_smooth * invE * d[c] / nd;
                const float gq = g[0] * q[0] + g[1] * q[1] + g[2] * q[2], hq = h[0] * q[0] + h[1] * q[1] + h[2] * q[2];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    out[c] += q[c] / (n + 1e-5f) - (n > 0.0f ? g[c] * gq / (n * (n + 1e-5f) * (n + 1e-5f)) : 0.0f);
                    outn[c] -= q[c] / (nh + 1e-5f) - (nh > 0.0f ? h[c] * hq / (nh * (nh + 1e-5f) * (nh + 1e-5f)) : 0.0f);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            L.g_theta[3 * t + c] = out[c];
            if (L.g_theta_nei) L.g_theta_nei[3 * t + c] = outn[c];
        }
    }
    double* outp = a.term_part + (size_t)blockIdx.x * NTERM;
#pragma unroll
    for (int k = 0; k < NTERM; ++k) {
        const double v = block_sum(term[k], red);
        if (threadIdx.x == 0) outp[k] = v;
    }
}

// terms[0..6] = rgb, eikonal, smooth, depth, gt_depth, normal_l1, normal_cos (unweighted, as SLAMLoss.get_* return them),
// terms[7] = their weighted sum
// one wave: lane j adds the partials of blocks j, j + 64, ... (in that order), then a fixed butterfly adds the 64 lane sums --
// deterministic, and ~700 dependent loads (a single thread took 84 us at the mapping shape) become 11 per lane
__global__ __launch_bounds__(64) void k_loss_final(LossArgs a) {
    const nsa_loss_t& L = a.in;
    double s[NTERM] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t k = threadIdx.x; k < a.term_blocks; k += 64)
        for (int j = 0; j < NTERM; ++j) s[j] += a.term_part[(size_t)k * NTERM + j];
    double M_all = 0.0, cnt_gt = 0.0;
    for (uint32_t i = threadIdx.x; i < L.bs * a.blocks_per_image; i += 64) {
        M_all += a.stat_part[(size_t)i * NSTAT + 2];
        cnt_gt += a.stat_part[(size_t)i * NSTAT + 6];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int j = 0; j < NTERM; ++j) s[j] += __shfl_xor(s[j], off);
        M_all += __shfl_xor(M_all, off);
        cnt_gt += __shfl_xor(cnt_gt, off);
    }
    if (threadIdx.x != 0) return;
    const double R = (double)L.bs * L.n;
    const float rgb = (float)(s[0] / (3.0 * R));
    const float eik = L.w_eik > 0.0f && L.E ? (float)(s[1] / L.E) : 0.0f;
    const float smooth = L.w_smooth > 0.0f && L.E && L.grad_theta_nei ? (float)(s[2] / L.E) : 0.0f;
    float depth = 0.0f;
    if (L.w_depth > 0.0f) {
        const float data = M_all != 0.0 ? (float)(s[3] / (2.0 * M_all)) : 0.0f;
        const float reg = M_all != 0.0 ? (float)(s[4] / M_all) : 0.0f;
        depth = data + 0.5f * reg;
    }
    const float gtd = L.w_gtdepth > 0.0f ? (float)(s[5] / cnt_gt) : 0.0f;          // 0/0 = NaN like an empty mean
    const float nl1 = (float)(s[6] / R), ncos = (float)(s[7] / R);
    float* t = L.terms;
    t[0] = rgb; t[1] = eik; t[2] = smooth; t[3] = depth; t[4] = gtd; t[5] = nl1; t[6] = ncos;
    t[7] = L.w_rgb * rgb + L.w_eik * eik + L.w_smooth * smooth + L.w_depth * depth + L.w_gtdepth * gtd + L.w_nl1 * nl1 + L.w_ncos * ncos;
}

static inline uint32_t loss_blocks_per_image(uint32_t n) {
    const uint32_t b = (n + LT - 1) / LT;
    return b < 1 ? 1 : (b > 16 ? 16 : b);
}

}  // namespace nsa

extern "C" uint64_t nsa_slam_loss_workspace(uint32_t bs, uint32_t n, uint32_t E) {
    using namespace nsa;
    const uint64_t R = (uint64_t)bs * n, T = R > E ? R : E;
    const uint64_t term_blocks = (T + LT - 1) / LT;
    return (uint64_t)bs * loss_blocks_per_image(n) * NSTAT * 2 + term_blocks * NTERM * 2 + R;      // in floats (doubles count twice)
}

extern "C" int nsa_slam_loss(const nsa_loss_t* in, float* workspace, nsa_stream_t stream) {
    using namespace nsa;
    if (!in || !workspace) return NSA_EBADARG;
    const nsa_loss_t& L = *in;
    if (!L.bs || !L.n || !L.S || !L.rgb || !L.rgb_gt || !L.depth || !L.depth_mono || !L.depth_real || !L.depth_real_mask ||
        !L.mask_gt || !L.sdf || !L.normal || !L.normal_gt || !L.g_rgb || !L.g_depth || !L.g_normal || !L.terms)
        return NSA_EBADARG;
    if (L.E && (!L.grad_theta || !L.g_theta || (L.grad_theta_nei && !L.g_theta_nei))) return NSA_EBADARG;
    LossArgs a;
    a.in = L;
    const uint64_t R = (uint64_t)L.bs * L.n, T = R > L.E ? R : L.E;
    a.blocks_per_image = loss_blocks_per_image(L.n);
    a.term_blocks = (uint32_t)((T + LT - 1) / LT);
    a.stat_part = reinterpret_cast<double*>(workspace);                                  // workspace must be 8-byte aligned
    a.term_part = a.stat_part + (size_t)L.bs * a.blocks_per_image * NSTAT;
    a.fg = reinterpret_cast<float*>(a.term_part + (size_t)a.term_blocks * NTERM);
    if (reinterpret_cast<uintptr_t>(workspace) & 7u) return NSA_EBADARG;
    launch_begin();
    hipLaunchKernelGGL(k_loss_stats, dim3(a.blocks_per_image, L.bs), dim3(LT), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_loss_terms, dim3(a.term_blocks), dim3(LT), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    return launch_end();
}
