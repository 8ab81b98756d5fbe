// render_composite.hip -- per-ray SDF -> density -> alpha compositing and its backward.
// Reference: SLAMNetwork.volume_rendering (code/model/network.py:349-370), the composite sums of SLAMNetwork.forward
// (:147-151, 281-300, 338-345) and GridPredefineDensity (code/model/density.py:37-67).
//
// One wave per ray (4 rays per 256-thread workgroup).  Lane l owns the `per = ceil(S/64)` consecutive samples
// l*per .. l*per+per-1, so the transmittance is a short serial prefix inside the lane plus one wave-shuffle
// exclusive scan of the lanes' free-energy sums; the ray sums (rgb, depth, normal, entropy) are wave reductions.
// Nothing goes through LDS or global scratch.
#include "sdf_net.hpp"

namespace nsa {

constexpr int MAX_PER = 4;   // S <= 256

__device__ __forceinline__ float beta_at(const float* __restrict__ voxels, uint32_t res, const float (&x)[3]) {
    const bool outside = fabsf(x[0]) > 0.99f || fabsf(x[1]) > 0.99f || fabsf(x[2]) > 0.99f;   // density.py:45-49
    float count = 0.0f;
    if (!outside) {
        int idx[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int v = (int)((x[k] + 1.0f) / 2.0f * (float)res);
            idx[k] = v < 0 ? 0 : (v >= (int)res ? (int)res - 1 : v);
        }
        count = voxels[((size_t)idx[0] * res + idx[1]) * res + idx[2]];
    }
    return 0.01207724805f * expf(-0.0116544676f * 0.0001f * count * 5.37538f) + 0.0023639156f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ float wave_excl(float v, int lane, float& total) {
    float incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float n = __shfl_up(incl, off);
        if (lane >= off) incl += n;
    }
    total = __shfl(incl, 63);
    // exclusive = inclusive of the previous lane (NOT incl - v: the last sample's free energy is ~1e10 * sigma and
    // the subtraction would cancel the whole prefix)
    const float prev = __shfl_up(incl, 1);
    return lane == 0 ? 0.0f : prev;
}

struct CompositeArgs {
    const float* rays_o;   // [R,3]
    const float* rays_d;   // [R,3]
    const float* z_vals;   // [R,S]
    const float* sdf;      // [R,S]
    const float* rgb;      // [R,S,3]
    const float* grad;     // [R,S,3]
    const float* voxels;
    uint32_t voxel_res, R, S;
    // forward outputs
    float* weights;        // [R,S]
    float* rgb_values;     // [R,3]
    float* depth;          // [R]   sum w z / (sum w + 1e-8)
    float* nmap;           // [R,3] sum w n, n = g / (|g| + 1e-6), world frame
    float* entropy;        // [R]   sum -w log(w + 1e-4)
    // backward inputs (per ray; any may be nullptr = zero) and outputs (per sample)
    const float* g_rgbv;   // [R,3]
    const float* g_depth;  // [R]
    const float* g_nmap;   // [R,3]
    const float* g_ent;    // [R]
    const float* g_w;      // [R,S]
    float* g_sdf;          // [R,S]
    float* g_rgb;          // [R,S,3]
    float* g_grad;         // [R,S,3]
};

struct Sample {
    float z, dist, sdf, beta, sigma, en, w, T;
};

// Shared forward recomputation: fills the lane's samples, returns the number it owns.
__device__ __forceinline__ int ray_forward(const CompositeArgs& a, uint32_t ray, int lane, uint32_t per, Sample (&sm)[MAX_PER]) {
    const uint32_t S = a.S;
    float o[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[k] = a.rays_o[ray * 3 + k]; d[k] = a.rays_d[ray * 3 + k]; }
    const float* zr = a.z_vals + (size_t)ray * S;
    const float* sr = a.sdf + (size_t)ray * S;
    const uint32_t i0 = lane * per;
    float esum = 0.0f;
    int n = 0;
#pragma unroll
    for (int k = 0; k < MAX_PER; ++k) {
        const uint32_t i = i0 + k;
        if ((uint32_t)k < per && i < S) {
            Sample& s = sm[k];
            s.z = zr[i];
            s.dist = i + 1 < S ? zr[i + 1] - s.z : 1e10f;            // network.py:356-357
            s.sdf = sr[i];
            float x[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) x[c] = o[c] + mul_rn(s.z, d[c]);
            s.beta = beta_at(a.voxels, a.voxel_res, x);
            const float sg = s.sdf > 0.0f ? 1.0f : (s.sdf < 0.0f ? -1.0f : 0.0f);
            s.sigma = (1.0f / s.beta) * (0.5f + 0.5f * sg * expm1f(-fabsf(s.sdf) / s.beta));
            s.en = s.dist * s.sigma;
            esum += s.en;
            n = k + 1;
        }
    }
    float tot;
    float run = wave_excl(esum, lane, tot);
#pragma unroll
    for (int k = 0; k < MAX_PER; ++k) {
        if (k < n) {
            Sample& s = sm[k];
            s.T = expf(-run);
            s.w = (1.0f - expf(-s.en)) * s.T;                          // :362-368
            run += s.en;
        }
    }
    return n;
}

#ifndef NSA_COMPOSITE_AS_HEADER   // (the kernels are defined once, in this translation unit)
__global__ __launch_bounds__(256) void k_composite_fwd(CompositeArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= a.R) return;
#include "composite_fwd_body.inc"
}

__global__ __launch_bounds__(256) void k_composite_bwd(CompositeArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= a.R) return;
    const uint32_t S = a.S, per = (S + 63) / 64;
    Sample sm[MAX_PER];
    const int n = ray_forward(a, ray, lane, per, sm);
    float grv[3] = {0, 0, 0}, gnm[3] = {0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (a.g_rgbv) grv[c] = a.g_rgbv[ray * 3 + c];
        if (a.g_nmap) gnm[c] = a.g_nmap[ray * 3 + c];
    }
    const float gdep = a.g_depth ? a.g_depth[ray] : 0.0f;
    const float gent = a.g_ent ? a.g_ent[ray] : 0.0f;
    // ray sums needed by the depth quotient
    float A = 0.0f, B = 0.0f;
#pragma unroll
    for (int k = 0; k < MAX_PER; ++k)
        if (k < n) { A = fmaf(sm[k].w, sm[k].z, A); B += sm[k].w; }
    A = wave_sum(A);
    B = wave_sum(B) + 1e-8f;
    // wbar_i and the per-sample colour / normal cotangents
    float wbar[MAX_PER], ww = 0.0f;
#pragma unroll
    for (int k = 0; k < MAX_PER; ++k) {
        wbar[k] = 0.0f;
        if (k < n) {
            const size_t i = (size_t)ray * S + lane * per + k;
            const Sample& s = sm[k];
            float g[3], c3[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { c3[c] = a.rgb[i * 3 + c]; g[c] = a.grad[i * 3 + c]; }
            const float nrm = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
            const float inv = 1.0f / (nrm + 1e-6f);
            float wb = grv[0] * c3[0] + grv[1] * c3[1] + grv[2] * c3[2];
            wb += gdep * (s.z * B - A) / (B * B);
            wb += (gnm[0] * g[0] + gnm[1] * g[1] + gnm[2] * g[2]) * inv;
            wb += gent * (-logf(s.w + 1e-4f) - s.w / (s.w + 1e-4f));
            if (a.g_w) wb += a.g_w[i];
            wbar[k] = wb;
            ww += wb * s.w;
            // c_bar = w * g_rgb ;  n = g/(|g|+eps): g_bar = nb/(|g|+eps) - g (g.nb) / (|g| (|g|+eps)^2), nb = w * g_nmap
            const float gdot = (g[0] * gnm[0] + g[1] * gnm[1] + g[2] * gnm[2]) * s.w;
            const float k2 = nrm > 0.0f ? gdot * inv * inv / nrm : 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                a.g_rgb[i * 3 + c] = s.w * grv[c];
                a.g_grad[i * 3 + c] = s.w * gnm[c] * inv - g[c] * k2;
            }
        }
    }
    // E_bar_k = wbar_k T_{k+1} - sum_{i>k} wbar_i w_i   (w_i = T_i - T_{i+1}).  The suffix sum is formed as a true
    // suffix scan (later lanes, then later samples of this lane), NOT as total - prefix: sigma_bar = E_bar * delta and
    // the last delta is 1e10, so a 1e-9 rounding residue in "total - prefix" would come out as a spurious gradient where
    // the reference's is exactly zero (the last sample has no later samples: the sum is empty).
    float later = ww;                                  // inclusive suffix over lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float n2 = __shfl_down(later, off);
        if (lane + off < 64) later += n2;
    }
    float suffix = __shfl_down(later, 1);              // exclusive: lanes after this one
    if (lane == 63) suffix = 0.0f;
#pragma unroll
    for (int k = MAX_PER - 1; k >= 0; --k) {
        if (k < n) {
            const size_t i = (size_t)ray * S + lane * per + k;
            const Sample& s = sm[k];
            const float Tn = s.T * expf(-s.en);                         // T_{k+1}
            const float eb = wbar[k] * Tn - suffix;
            suffix += wbar[k] * s.w;
            const float sb = eb * s.dist;                               // sigma_bar
            const float sg = s.sdf > 0.0f ? 1.0f : (s.sdf < 0.0f ? -1.0f : 0.0f);
            // d sigma / d sdf as torch's autograd forms it: the backward of expm1 is (result + 1), NOT exp(x) -- far from the
            // surface (|s|/beta > ~17) expm1 has rounded to exactly -1 and the reference's derivative is exactly 0, while
            // exp(-|s|/beta) is 1e-8 .. 1e-16 and the last interval multiplies it by 1e10 (density.py:37-39, network.py:357)
            const float em1 = expm1f(-fabsf(s.sdf) / s.beta) + 1.0f;
            const float dsig = -0.5f * sg * sg * em1 / (s.beta * s.beta);
            a.g_sdf[i] = sb * dsig;
        }
    }
}

// Tracking objective in one pass (graph-captured tracker): composite forward for the rendered colour, d mean|rgb - gt| / d rgb
// (loss.py:57-65,131 with torch.nn.L1Loss) and the composite backward of that cotangent alone -- the arithmetic of
// k_composite_fwd -> k_l1_loss -> k_composite_bwd(g_rgbv only) term for term, with the forward state in registers instead of
// three launches that each recompute it.  ray_loss[r] = sum_c |rgb_c - gt_c|; the caller adds the rays in a fixed order.
__global__ __launch_bounds__(256) void k_composite_track(CompositeArgs a, const float* __restrict__ gt, float* __restrict__ ray_loss,
                                                         float inv_n) {
    const int lane = threadIdx.x & 63;
    const uint32_t ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= a.R) return;
#include "composite_track_body.inc"
}

// g_o[r] = sum_i g_x[r,i] ; g_d[r] = sum_i z_i g_x[r,i] + sum_i g_dir[r,i]        (x = o + z d, view dir = d)
__global__ __launch_bounds__(256) void k_rays_bwd(const float* __restrict__ z_vals, const float* __restrict__ g_x,
                                                  const float* __restrict__ g_dir, float* __restrict__ g_o,
                                                  float* __restrict__ g_d, uint32_t R, uint32_t S) {
    const int lane = threadIdx.x & 63;
    const uint32_t ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= R) return;
    float acc[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i = lane; i < S; i += 64) {
        const size_t p = (size_t)ray * S + i;
        const float z = z_vals[p];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float gx = g_x[p * 3 + c];
            acc[c] += gx;
            acc[3 + c] += z * gx + (g_dir ? g_dir[p * 3 + c] : 0.0f);
        }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) acc[q] = wave_sum(acc[q]);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { g_o[ray * 3 + c] = acc[c]; g_d[ray * 3 + c] = acc[3 + c]; }
    }
}

#endif  // NSA_COMPOSITE_AS_HEADER

}  // namespace nsa

#ifndef NSA_COMPOSITE_AS_HEADER   // (render_colour.hip includes this file for the kernel pieces only: k_colour_fwd_track)
extern "C" {

int nsa_composite_forward(const float* rays_o, const float* rays_d, const float* z_vals, const float* sdf, const float* rgb,
                          const float* grad, const float* voxels, uint32_t voxel_res, uint32_t R, uint32_t S, float* weights,
                          float* rgb_values, float* depth, float* nmap, float* entropy, nsa_stream_t stream) {
    using namespace nsa;
    if (R == 0) return NSA_OK;
    if (!rays_o || !rays_d || !z_vals || !sdf || !rgb || !grad || !voxels || !weights || !rgb_values || !depth || !nmap ||
        !entropy || S == 0 || S > 64 * MAX_PER)
        return NSA_EBADARG;
    CompositeArgs a{};
    a.rays_o = rays_o; a.rays_d = rays_d; a.z_vals = z_vals; a.sdf = sdf; a.rgb = rgb; a.grad = grad; a.voxels = voxels;
    a.voxel_res = voxel_res; a.R = R; a.S = S;
    a.weights = weights; a.rgb_values = rgb_values; a.depth = depth; a.nmap = nmap; a.entropy = entropy;
    launch_begin();
    hipLaunchKernelGGL(k_composite_fwd, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    return launch_end();
}

int nsa_composite_backward(const float* rays_o, const float* rays_d, const float* z_vals, const float* sdf, const float* rgb,
                           const float* grad, const float* voxels, uint32_t voxel_res, uint32_t R, uint32_t S,
                           const float* g_rgb_values, const float* g_depth, const float* g_nmap, const float* g_entropy,
                           const float* g_weights, float* g_sdf, float* g_rgb, float* g_grad, nsa_stream_t stream) {
    using namespace nsa;
    if (R == 0) return NSA_OK;
    if (!rays_o || !rays_d || !z_vals || !sdf || !rgb || !grad || !voxels || !g_sdf || !g_rgb || !g_grad || S == 0 ||
        S > 64 * MAX_PER)
        return NSA_EBADARG;
    CompositeArgs a{};
    a.rays_o = rays_o; a.rays_d = rays_d; a.z_vals = z_vals; a.sdf = sdf; a.rgb = rgb; a.grad = grad; a.voxels = voxels;
    a.voxel_res = voxel_res; a.R = R; a.S = S;
    a.g_rgbv = g_rgb_values; a.g_depth = g_depth; a.g_nmap = g_nmap; a.g_ent = g_entropy; a.g_w = g_weights;
    a.g_sdf = g_sdf; a.g_rgb = g_rgb; a.g_grad = g_grad;
    launch_begin();
    hipLaunchKernelGGL(k_composite_bwd, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    return launch_end();
}

int nsa_composite_track(const float* rays_o, const float* rays_d, const float* z_vals, const float* sdf, const float* rgb,
                        const float* voxels, uint32_t voxel_res, uint32_t R, uint32_t S, const float* gt, uint32_t n_total,
                        float* rgb_values, float* ray_loss, float* g_sdf, float* g_rgb, float* g_grad, nsa_stream_t stream) {
    using namespace nsa;
    if (R == 0) return NSA_OK;
    if (!rays_o || !rays_d || !z_vals || !sdf || !rgb || !voxels || !gt || !rgb_values || !ray_loss || !g_sdf || !g_rgb ||
        !g_grad || S == 0 || S > 64 * MAX_PER || n_total < R)
        return NSA_EBADARG;
    CompositeArgs a{};
    a.rays_o = rays_o; a.rays_d = rays_d; a.z_vals = z_vals; a.sdf = sdf; a.rgb = rgb; a.voxels = voxels;
    a.voxel_res = voxel_res; a.R = R; a.S = S;
    a.rgb_values = rgb_values; a.g_sdf = g_sdf; a.g_rgb = g_rgb; a.g_grad = g_grad;
    launch_begin();
    hipLaunchKernelGGL(k_composite_track, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, gt, ray_loss,
                       1.0f / (float)(3 * (uint64_t)n_total));
    return launch_end();
}

int nsa_rays_backward(const float* z_vals, const float* g_x, const float* g_dir, uint32_t R, uint32_t S, float* g_rays_o,
                      float* g_rays_d, nsa_stream_t stream) {
    using namespace nsa;
    if (R == 0) return NSA_OK;
    if (!z_vals || !g_x || !g_rays_o || !g_rays_d || S == 0) return NSA_EBADARG;
    launch_begin();
    hipLaunchKernelGGL(k_rays_bwd, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, z_vals, g_x, g_dir, g_rays_o,
                       g_rays_d, R, S);
    return launch_end();
}

}  // extern "C"
#endif  // NSA_COMPOSITE_AS_HEADER
