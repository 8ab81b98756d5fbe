// render_rays.hip -- pixel -> ray lifting and its backward to the camera-to-world matrix.
// Reference: rend_util.get_camera_params + lift (code/utils/rend_util.py:68-93,107-129) and the identity-pose second
// call that yields depth_scale (code/model/network.py:99-102).  One thread per ray; the backward reduces
// d/d(pose[:3,:3]) = sum_rays vbar c^T and d/d(pose[:3,3]) = sum_rays obar per image in a fixed order (no atomics).
#include "grid_common.hpp"
#include "draw_common.hpp"

namespace nsa {

struct RaysArgs {
    const float* uv;     // [b,n,2]
    const float* pose;   // [b,4,4] camera-to-world
    const float* K;      // [b,4,4]
    uint32_t b, n;
    float* rays_o;       // [b*n,3]
    float* rays_d;       // [b*n,3]   (p - o) / |p - o|^2  -- NOT unit length (rend_util.py:92)
    float* depth_scale;  // [b*n]     z component of the identity-pose ray
    const float* g_o;    // [b*n,3]
    const float* g_d;    // [b*n,3]
    float* g_pose;       // [b,4,4] overwritten
};

__device__ __forceinline__ void lift_pixel(const float* __restrict__ K, float u, float v, float (&c)[3]) {
    const float fx = K[0], sk = K[1], cx = K[2], fy = K[5], cy = K[6];
    c[0] = (u - cx + cy * sk / fy - sk * v / fy) / fx;       // rend_util.py:117-125 (z = 1)
    c[1] = (v - cy) / fy;
    c[2] = 1.0f;
}

__device__ __forceinline__ void rays_fwd_block(const RaysArgs& a, const uint32_t bid) {
    const uint32_t r = bid * 256 + threadIdx.x;
    if (r >= a.b * a.n) return;
    const uint32_t bi = r / a.n;
    const float* P = a.pose + bi * 16;
    float c[3];
    lift_pixel(a.K + bi * 16, a.uv[2 * r], a.uv[2 * r + 1], c);
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float w = P[4 * k] * c[0] + P[4 * k + 1] * c[1] + P[4 * k + 2] * c[2] + P[4 * k + 3];   // bmm with [x,y,1,1]
        v[k] = w - P[4 * k + 3];                                                                       // - cam_loc
        a.rays_o[3 * r + k] = P[4 * k + 3];
    }
    const float s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) a.rays_d[3 * r + k] = v[k] / s;
    a.depth_scale[r] = c[2] / (c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
}

__global__ __launch_bounds__(256) void k_rays_fwd(RaysArgs a) { rays_fwd_block(a, blockIdx.x); }

// the same launch also makes the sampler's random draws of the pass (draw_common.hpp): workgroups [0, ray_blocks) lift the rays, the
// others are nsa_draw's rand / pick workgroups -- one graph node less in front of the sampler
__global__ __launch_bounds__(256) void k_rays_fwd_draw(RaysArgs a, DrawArgs d, uint32_t ray_blocks) {
    if (blockIdx.x < ray_blocks) rays_fwd_block(a, blockIdx.x);
    else draw_block(d, blockIdx.x - ray_blocks, gridDim.x - ray_blocks);
}

// One workgroup per image, every sum in a fixed order (thread t adds rays t, t + 1024, ...; butterfly over the lanes; waves in wave
// order): the camera gradient of the eager autograd path is reproducible to the bit, like the kernel tracker's (track_tail.hip).
// (Until round 4: one thread per ray and 12 atomics per wave -- run-to-run differences in the last bits of the pose gradient, which
// an optimizer step turns into last-bit differences of every later iteration.)
constexpr int POSE_BWD_W = 16;
__global__ __launch_bounds__(64 * POSE_BWD_W) void k_rays_pose_bwd(RaysArgs a) {
    __shared__ float part[POSE_BWD_W][12];
    const uint32_t bi = blockIdx.x;
    const float* P = a.pose + bi * 16;
    const float* Kb = a.K + bi * 16;
    float acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = 0.0f;
    for (uint32_t i = threadIdx.x; i < a.n; i += 64 * POSE_BWD_W) {
        const uint32_t r = bi * a.n + i;
        float c[3];
        lift_pixel(Kb, a.uv[2 * r], a.uv[2 * r + 1], c);
        float v[3], gd[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float w = P[4 * k] * c[0] + P[4 * k + 1] * c[1] + P[4 * k + 2] * c[2] + P[4 * k + 3];
            v[k] = w - P[4 * k + 3];
            gd[k] = a.g_d[3 * r + k];
        }
        const float s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
        const float vg = v[0] * gd[0] + v[1] * gd[1] + v[2] * gd[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float vb = gd[k] / s - 2.0f * v[k] * vg / (s * s);     // d = v / (v.v)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[4 * k + j] += vb * c[j];
            acc[4 * k + 3] += a.g_o[3 * r + k];                          // cam_loc = pose[:3,3]; (w - cam_loc) cancels
        }
    }
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        float x = acc[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if (lane == 0) part[wv][q] = x;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        float g = 0.0f;
        if (threadIdx.x < 12) {
#pragma unroll
            for (int w = 0; w < POSE_BWD_W; ++w) g += part[w][threadIdx.x];
        }
        a.g_pose[bi * 16 + threadIdx.x] = g;                              // bottom row: zero
    }
}

}  // namespace nsa

extern "C" {

int nsa_rays_forward(const float* uv, const float* pose, const float* K, uint32_t b, uint32_t n, float* rays_o, float* rays_d,
                     float* depth_scale, nsa_stream_t stream) {
    using namespace nsa;
    if (b * n == 0) return NSA_OK;
    if (!uv || !pose || !K || !rays_o || !rays_d || !depth_scale) return NSA_EBADARG;
    RaysArgs a{uv, pose, K, b, n, rays_o, rays_d, depth_scale, nullptr, nullptr, nullptr};
    launch_begin();
    hipLaunchKernelGGL(k_rays_fwd, dim3((b * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return launch_end();
}

int nsa_rays_forward_draw(const float* uv, const float* pose, const float* K, uint32_t b, uint32_t n, float* rays_o, float* rays_d,
                          float* depth_scale, uint64_t* state, uint64_t n_rand, float* t_rand, uint32_t E, uint32_t n_extra, uint32_t S,
                          int32_t* extra_idx, nsa_stream_t stream) {
    using namespace nsa;
    if (b * n == 0) return NSA_EBADARG;
    if (!uv || !pose || !K || !rays_o || !rays_d || !depth_scale) return NSA_EBADARG;
    if (!draw_args_ok(state, n_rand, t_rand, E, n_extra, b * n, S, extra_idx, nullptr)) return NSA_EBADARG;
    RaysArgs a{uv, pose, K, b, n, rays_o, rays_d, depth_scale, nullptr, nullptr, nullptr};
    DrawArgs d{};
    const uint32_t draw_blocks = draw_launch_shape(d, reinterpret_cast<unsigned long long*>(state), n_rand, t_rand, E, n_extra, b * n, S,
                                                   extra_idx, nullptr);
    const uint32_t ray_blocks = (b * n + 255) / 256;
    launch_begin();
    hipLaunchKernelGGL(k_rays_fwd_draw, dim3(ray_blocks + draw_blocks), dim3(256), 0, (hipStream_t)stream, a, d, ray_blocks);
    return launch_end();
}

int nsa_rays_pose_backward(const float* uv, const float* pose, const float* K, uint32_t b, uint32_t n, const float* g_rays_o,
                           const float* g_rays_d, float* g_pose, nsa_stream_t stream) {
    using namespace nsa;
    if (!g_pose) return NSA_EBADARG;
    launch_begin();
    if (b == 0) return NSA_OK;
    if (n == 0) return hipMemsetAsync(g_pose, 0, sizeof(float) * 16 * b, (hipStream_t)stream) == hipSuccess ? NSA_OK : NSA_ELAUNCH;
    if (!uv || !pose || !K || !g_rays_o || !g_rays_d) return NSA_EBADARG;
    RaysArgs a{uv, pose, K, b, n, nullptr, nullptr, nullptr, g_rays_o, g_rays_d, g_pose};
    hipLaunchKernelGGL(k_rays_pose_bwd, dim3(b), dim3(64 * POSE_BWD_W), 0, (hipStream_t)stream, a);
    return launch_end();
}

}  // extern "C"
