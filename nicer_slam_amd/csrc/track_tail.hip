// track_tail.hip -- the scalar head and tail of a camera-tracking iteration, so that a whole iteration is a fixed
// sequence of our kernels with no autograd bookkeeping in between (SURVEY 8a rows a16, a17 + the camera optimizer):
//   k_cam_to_pose        7-vector (qw,qx,qy,qz,tx,ty,tz) -> 4x4 camera-to-world, two_s = 2/|q|^2
//                        (code/utils/general.py:52-100: quad2rotation / get_camera_from_tensor)
//   k_l1_loss            loss = mean |rgb - gt| and d loss / d rgb = sign(rgb - gt) / N
//                        (code/model/loss.py:57-65,131 with rgb_loss = torch.nn.L1Loss, the tracking objective)
//   k_pose_grad_to_cam   backward of k_cam_to_pose
//   k_adam               torch.optim.Adam update of the (tiny) camera vector, optional StepLR schedule
//                        (code/training/volsdf_train.py:396-399,425-427)
#include "grid_common.hpp"
#include "draw_common.hpp"

namespace nsa {

__device__ __forceinline__ void cam_to_pose(const float* __restrict__ q, float* __restrict__ P) {
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float s = 2.0f / (r * r + i * i + j * j + k * k);
    P[0] = -s * (j * j + k * k) + 1.0f;  P[1] = s * (i * j - k * r);          P[2] = s * (i * k + j * r);           P[3] = q[4];
    P[4] = s * (i * j + k * r);          P[5] = -s * (i * i + k * k) + 1.0f;  P[6] = s * (j * k - i * r);           P[7] = q[5];
    P[8] = s * (i * k - j * r);          P[9] = s * (j * k + i * r);          P[10] = -s * (i * i + j * j) + 1.0f;  P[11] = q[6];
    P[12] = 0.0f; P[13] = 0.0f; P[14] = 0.0f; P[15] = 1.0f;
}

__global__ void k_cam_to_pose(const float* __restrict__ cam, float* __restrict__ pose, uint32_t b) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= b) return;
    cam_to_pose(cam + 7 * n, pose + 16 * n);
}

// o[0..6] = d loss / d cam from G = d loss / d pose (row-major 4x4; rows 0..2 used)
__device__ __forceinline__ void pose_grad_to_cam(const float* __restrict__ q, const float* __restrict__ G, float* __restrict__ o) {
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float nn = r * r + i * i + j * j + k * k;
    const float s = 2.0f / nn;
    const float N[9] = {-(j * j + k * k), i * j - k * r, i * k + j * r, i * j + k * r, -(i * i + k * k), j * k - i * r,
                        i * k - j * r, j * k + i * r, -(i * i + j * j)};
    float A = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) A += G[4 * a + c] * N[3 * a + c];
#define GG(a, c) G[4 * (a) + (c)]
    const float dr = -k * GG(0, 1) + j * GG(0, 2) + k * GG(1, 0) - i * GG(1, 2) - j * GG(2, 0) + i * GG(2, 1);
    const float di = j * GG(0, 1) + k * GG(0, 2) + j * GG(1, 0) - 2 * i * GG(1, 1) - r * GG(1, 2) + k * GG(2, 0) + r * GG(2, 1) - 2 * i * GG(2, 2);
    const float dj = -2 * j * GG(0, 0) + i * GG(0, 1) + r * GG(0, 2) + i * GG(1, 0) + k * GG(1, 2) - r * GG(2, 0) + k * GG(2, 1) - 2 * j * GG(2, 2);
    const float dk = -2 * k * GG(0, 0) - r * GG(0, 1) + i * GG(0, 2) + r * GG(1, 0) - 2 * k * GG(1, 1) + j * GG(1, 2) + i * GG(2, 0) + j * GG(2, 1);
    o[0] = -s * s * r * A + s * dr;
    o[1] = -s * s * i * A + s * di;
    o[2] = -s * s * j * A + s * dj;
    o[3] = -s * s * k * A + s * dk;
    o[4] = GG(0, 3); o[5] = GG(1, 3); o[6] = GG(2, 3);
#undef GG
}

// out[0..6] = d loss / d cam of image n (n < b), from g_pose[b,4,4]
__global__ void k_pose_grad_to_cam(const float* __restrict__ cam, const float* __restrict__ g_pose, float* __restrict__ g_cam,
                                   uint32_t b) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= b) return;
    pose_grad_to_cam(cam + 7 * n, g_pose + 16 * n, g_cam + 7 * n);
}

// single block; n = number of scalars (3 R)
__global__ __launch_bounds__(256) void k_l1_loss(const float* __restrict__ pred, const float* __restrict__ target, float* __restrict__ loss,
                                                 float* __restrict__ g_pred, uint32_t n) {
    __shared__ float part[4];
    const float inv = 1.0f / (float)n;
    float acc = 0.0f;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const float d = pred[i] - target[i];
        acc += fabsf(d);
        g_pred[i] = d > 0.0f ? inv : (d < 0.0f ? -inv : 0.0f);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (part[0] + part[1] + part[2] + part[3]) * inv;
}

struct AdamArgs {
    float* p; const float* g; float* m; float* v; float* step;   // step: device scalar, incremented here
    const float* g_div;                                          // optional device scalar: the gradient is g / g_div[0]
    uint32_t n;
    float lr, beta1, beta2, eps, lr_gamma;
    uint32_t lr_step;                                            // 0 = constant lr
    // optional arg-min-loss camera of the frame (volsdf_train.py:402-403,441-446): best[0] = smallest loss so far,
    // best[1..n] = the parameters AFTER the step of the iteration that produced it (the reference clones camera_tensor
    // after optimizer_camera.step()); loss = this iteration's loss (divided by loss_div[0] when given)
    float* best; const float* loss; const float* loss_div;
};

// after the step: keep the camera of the smallest loss (strict <, like the reference); called by ONE thread
__device__ __forceinline__ void keep_best(const AdamArgs& a) {
    if (!a.best || !a.loss) return;
    const float l = a.loss_div ? a.loss[0] / a.loss_div[0] : a.loss[0];
    if (l < a.best[0]) {
        a.best[0] = l;
        for (uint32_t i = 0; i < a.n; ++i) a.best[1 + i] = a.p[i];
    }
}

// b^e for a non-negative integer e by squaring (<= 2 log2 e double multiplications, a few ulp of double: the bias corrections and
// the StepLR factor round to the same fp32 as Python's pow; the library pow costs ~400 fp64 instructions on a lone wave)
__device__ __forceinline__ double ipow(double b, double e_real) {
    unsigned long long e = (unsigned long long)e_real;
    double r = 1.0;
    while (e) {
        if (e & 1ull) r *= b;
        b *= b;
        e >>= 1;
    }
    return r;
}

// (m0, v0, p0: the moments and the parameter as they were BEFORE the step, g: the gradient -- loaded by the caller, possibly long
// before; returns the stepped parameter)
__device__ __forceinline__ float adam_one(const AdamArgs& a, uint32_t i, float t, float m0, float v0, float p0, float g) {
    {
        const float m = a.beta1 * m0 + (1.0f - a.beta1) * g;
        const float v = a.beta2 * v0 + (1.0f - a.beta2) * g * g;
        a.m[i] = m;
        a.v[i] = v;
        const double bc1 = 1.0 - ipow((double)a.beta1, (double)t);
        const double bc2 = 1.0 - ipow((double)a.beta2, (double)t);
        float lr = a.lr;
        if (a.lr_step) lr *= (float)ipow((double)a.lr_gamma, floor(((double)t - 1.0) / (double)a.lr_step));
        const float step_size = (float)((double)lr / bc1);
        const float denom = sqrtf(v) / (float)sqrt(bc2) + a.eps;
        const float p = p0 - step_size * m / denom;
        a.p[i] = p;
        return p;
    }
}
__device__ __forceinline__ void adam_one(const AdamArgs& a, uint32_t i, float t) {
    adam_one(a, i, t, a.m[i], a.v[i], a.p[i], a.g_div ? a.g[i] / a.g_div[0] : a.g[i]);
}

__global__ void k_adam(AdamArgs a) {
    const uint32_t i = threadIdx.x;
    const float t = a.step[0] + 1.0f;
    if (i < a.n) adam_one(a, i, t);
    __syncthreads();                        // (global writes of this block are visible to it after the barrier)
    if (i == 0) {
        a.step[0] = t;
        keep_best(a);
    }
}

// ---- fused head / tail of a single-image tracking iteration ------------------------------------------------------
// head: camera 7-vector -> pose (every thread, in registers) -> rays; also stores the pose for later consumers.
// tail: per-ray pose-gradient terms (backward of the ray lifting, see render_rays.hip::k_rays_pose_bwd), block
//       reduction, backward of cam->pose, and -- single GPU -- the Adam step: 5 graph nodes become 1.
struct TrackArgs {
    const float* uv;      // [n,2]
    const float* K;       // [4,4]
    float* cam;           // [7]
    float* pose;          // [4,4] out (head)
    uint32_t n;
    float* rays_o; float* rays_d; float* depth_scale;       // head outputs
    const float* g_o; const float* g_d;                       // tail inputs [n,3]
    float* g_cam;                                             // [7] out (tail); [9] with reduce_weight
    float reduce_weight;                                      // > 0 (multi-GPU): g_cam[0..7] *= w, g_cam[8] = w -- the
                                                              // buffer is then summed over ranks and divided by [8]
    AdamArgs adam;                                            // adam.p == nullptr: no step
};

__device__ __forceinline__ void lift_pixel_k(const float* __restrict__ K, float u, float v, float (&c)[3]) {
    const float fx = K[0], sk = K[1], cx = K[2], fy = K[5], cy = K[6];
    c[0] = (u - cx + cy * sk / fy - sk * v / fy) / fx;       // rend_util.py:117-125 (z = 1)
    c[1] = (v - cy) / fy;
    c[2] = 1.0f;
}

__global__ __launch_bounds__(256) void k_track_head(TrackArgs a) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    float P[16];
    cam_to_pose(a.cam, P);
    if (r == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a.pose[i] = P[i];
    }
    if (r >= a.n) return;
    float c[3];
    lift_pixel_k(a.K, a.uv[2 * r], a.uv[2 * r + 1], c);
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float w = P[4 * k] * c[0] + P[4 * k + 1] * c[1] + P[4 * k + 2] * c[2] + P[4 * k + 3];
        v[k] = w - P[4 * k + 3];
        a.rays_o[3 * r + k] = P[4 * k + 3];
    }
    const float s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) a.rays_d[3 * r + k] = v[k] / s;
    a.depth_scale[r] = c[2] / (c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
}

__global__ __launch_bounds__(1024) void k_track_tail(TrackArgs a) {
    __shared__ float part[16][12];
    __shared__ float G[16];
    float P[16];
    cam_to_pose(a.cam, P);
    float acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = 0.0f;
    for (uint32_t r = threadIdx.x; r < a.n; r += 1024) {
        float c[3];
        lift_pixel_k(a.K, a.uv[2 * r], a.uv[2 * r + 1], c);
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
            acc[4 * k + 3] += a.g_o[3 * r + k];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        float x = acc[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if (lane == 0) part[wave][q] = x;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        float x = 0.0f;
#pragma unroll
        for (int w = 0; w < 16; ++w) x += part[w][threadIdx.x];
        G[threadIdx.x] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float o[7];
        pose_grad_to_cam(a.cam, G, o);
#pragma unroll
        for (int i = 0; i < 7; ++i) a.g_cam[i] = o[i];
        if (a.reduce_weight > 0.0f) {       // slot 7 holds this rank's loss (k_l1_loss), slot 8 the weight
#pragma unroll
            for (int i = 0; i < 8; ++i) a.g_cam[i] *= a.reduce_weight;
            a.g_cam[8] = a.reduce_weight;
        }
    }
    __syncthreads();                        // g_cam is visible block-wide (global writes + barrier)
    if (a.adam.p) {                         // one thread per parameter (the double-precision pow calls run in parallel)
        const float t = a.adam.step[0] + 1.0f;
        if (threadIdx.x < 7) adam_one(a.adam, threadIdx.x, t);
        __syncthreads();
        if (threadIdx.x == 0) {
            a.adam.step[0] = t;
            keep_best(a.adam);
        }
    }
}

// ---- the same head and tail with the neighbouring per-ray work folded in (graph-captured tracker, one ray chunk) ---------------
// begin:  the frame's pixel batch is copied into the buffers the captured graph reads (uv, gt) by the kernel that lifts the
//         rays -- one eager launch in front of the replay instead of two copies + a graph node.
// finish: k_rays_bwd (wave per ray) + the per-ray pose-gradient terms of k_track_tail; block partials (fixed order) go to
//         `part`, and the LAST block to arrive (ticket) adds them in a fixed order, forms the loss from the rays' L1 sums and runs
//         the backward of cam -> pose and the Adam step: deterministic, one launch instead of two and no 1024-ray serial block.
struct FinishArgs {
    TrackArgs t;
    const float* z_vals;  // [n,S]
    const float* g_x;     // [n,S,3]
    const float* g_dir;   // [n,S,3]
    const float* ray_loss;  // [n] sum_c |rgb_c - gt_c| (k_composite_track)
    uint32_t S;
    float inv_n;          // 1 / (3 n): the loss is the mean over the pass's 3 n colour values
    float* part;          // [blocks][16] workspace
    unsigned* ticket;     // zero before the first launch; the kernel leaves it zero
};

constexpr int FIN_Q = 13;   // 12 pose-gradient entries (rows 0..2 of the 4x4) + the L1 sum

__device__ __forceinline__ void track_begin_block(const TrackArgs& a, const float* uv_in, const float* gt_in, float* uv, float* gt,
                                                  const uint32_t bid) {
    // (uv_in / gt_in may BE uv / gt: every thread reads its own elements before it writes them)
    const uint32_t r = bid * 256 + threadIdx.x;
    float P[16];
    cam_to_pose(a.cam, P);
    if (r == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a.pose[i] = P[i];
    }
    if (r >= a.n) return;
    const float pu = uv_in[2 * r], pv = uv_in[2 * r + 1];
    uv[2 * r] = pu;
    uv[2 * r + 1] = pv;
#pragma unroll
    for (int k = 0; k < 3; ++k) gt[3 * r + k] = gt_in[3 * r + k];
    float c[3];
    lift_pixel_k(a.K, pu, pv, c);
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float w = P[4 * k] * c[0] + P[4 * k + 1] * c[1] + P[4 * k + 2] * c[2] + P[4 * k + 3];
        v[k] = w - P[4 * k + 3];
        a.rays_o[3 * r + k] = P[4 * k + 3];
    }
    const float s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) a.rays_d[3 * r + k] = v[k] / s;
    a.depth_scale[r] = c[2] / (c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
}

__global__ __launch_bounds__(256) void k_track_begin(TrackArgs a, const float* uv_in, const float* gt_in, float* uv, float* gt) {
    track_begin_block(a, uv_in, gt_in, uv, gt, blockIdx.x);
}

// The same launch also makes the iteration's random draws (draw_common.hpp: what nsa_draw launches as a graph node of its own):
// workgroups [0, begin_blocks) lift the rays, the others are the draw's rand / pick workgroups.  The two halves share nothing.
__global__ __launch_bounds__(256) void k_track_begin_draw(TrackArgs a, const float* uv_in, const float* gt_in, float* uv, float* gt,
                                                          DrawArgs d, uint32_t begin_blocks) {
    if (blockIdx.x < begin_blocks) track_begin_block(a, uv_in, gt_in, uv, gt, blockIdx.x);
    else draw_block(d, blockIdx.x - begin_blocks, gridDim.x - begin_blocks);
}

// FIN_W rays per workgroup: the ticket is one device-scope atomic on ONE address per workgroup, and those serialise at the memory
// side (~15 ns each): 64 workgroups of 16 waves instead of 256 of 4.
constexpr int FIN_W = 16;

__global__ __launch_bounds__(64 * FIN_W) void k_track_finish(FinishArgs f) {
    __shared__ float rp[FIN_W][FIN_Q];
    __shared__ float G[16];
    __shared__ unsigned last;
    const TrackArgs& a = f.t;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t ray = blockIdx.x * FIN_W + wv;
    float P[16];
    cam_to_pose(a.cam, P);
    // optimizer state as it is before the step, requested now by every block: only the last block to arrive uses it, and there
    // these loads would otherwise sit at the end of a chain of dependent round trips (ticket -> partials -> gradient -> state)
    float t0 = 0.0f, m0 = 0.0f, v0 = 0.0f, p0 = 0.0f;
    if (a.adam.p && threadIdx.x < 7) {
        t0 = a.adam.step[0];
        m0 = a.adam.m[threadIdx.x];
        v0 = a.adam.v[threadIdx.x];
        p0 = a.adam.p[threadIdx.x];
    }
    const float best0 = (a.adam.p && a.adam.best && threadIdx.x == 0) ? a.adam.best[0] : 0.0f;
    float vals[FIN_Q];
#pragma unroll
    for (int q = 0; q < FIN_Q; ++q) vals[q] = 0.0f;
    if (ray < a.n) {
        float acc[6] = {0, 0, 0, 0, 0, 0};          // g_o, g_d of the ray: render_composite.hip::k_rays_bwd
        for (uint32_t i = lane; i < f.S; i += 64) {
            const size_t p = (size_t)ray * f.S + i;
            const float z = f.z_vals[p];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float gx = f.g_x[p * 3 + c];
                acc[c] += gx;
                acc[3 + c] += z * gx + f.g_dir[p * 3 + c];
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[q] += __shfl_xor(acc[q], off);
        }
        float c[3], v[3];
        lift_pixel_k(a.K, a.uv[2 * ray], a.uv[2 * ray + 1], c);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float w = P[4 * k] * c[0] + P[4 * k + 1] * c[1] + P[4 * k + 2] * c[2] + P[4 * k + 3];
            v[k] = w - P[4 * k + 3];
        }
        const float s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
        const float vg = v[0] * acc[3] + v[1] * acc[4] + v[2] * acc[5];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float vb = acc[3 + k] / s - 2.0f * v[k] * vg / (s * s);     // d = v / (v.v)
#pragma unroll
            for (int j = 0; j < 3; ++j) vals[4 * k + j] = vb * c[j];
            vals[4 * k + 3] = acc[k];
        }
        vals[12] = f.ray_loss[ray];
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < FIN_Q; ++q) rp[wv][q] = vals[q];
    }
    __syncthreads();
    if (wv == 0) {                          // one wave publishes the block's partial sums and takes the ticket
        if (lane < FIN_Q) {
            float x = 0.0f;
#pragma unroll
            for (int w = 0; w < FIN_W; ++w) x += rp[w][lane];
            __hip_atomic_store(f.part + (size_t)blockIdx.x * 16 + lane, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __threadfence();                    // the partials are visible device-wide before the ticket is taken
        if (lane == 0) last = atomicAdd(f.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // the last block adds the partials in a fixed order: thread t takes blocks t, t + 1024, ... (all 13 loads of a block row are
    // independent and in flight together), then a butterfly over the 64 lanes and the waves in wave order
    {
        float x[FIN_Q];
#pragma unroll
        for (int q = 0; q < FIN_Q; ++q) x[q] = 0.0f;
        for (uint32_t b = threadIdx.x; b < gridDim.x; b += 64 * FIN_W) {
            float v[FIN_Q];
#pragma unroll
            for (int q = 0; q < FIN_Q; ++q)
                v[q] = __hip_atomic_load(f.part + (size_t)b * 16 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 0; q < FIN_Q; ++q) x[q] += v[q];
        }
#pragma unroll
        for (int q = 0; q < FIN_Q; ++q) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) x[q] += __shfl_xor(x[q], off);
        }
        __syncthreads();                    // rp is free again (every wave passed the read above before the ticket barrier)
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < FIN_Q; ++q) rp[wv][q] = x[q];
        }
        __syncthreads();
        if (threadIdx.x < FIN_Q) {
            float g = 0.0f;
#pragma unroll
            for (int w = 0; w < FIN_W; ++w) g += rp[w][threadIdx.x];
            G[threadIdx.x] = g;
        }
    }
    __syncthreads();
    __shared__ float msg[9], stepped[8];    // the message as it goes to global memory; the camera after the step
    if (threadIdx.x == 0) {
        f.ticket[0] = 0u;
        float o[8];
        pose_grad_to_cam(a.cam, G, o);
        o[7] = G[12] * f.inv_n;
        const float w = a.reduce_weight > 0.0f ? a.reduce_weight : 1.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { msg[i] = o[i] * w; a.g_cam[i] = msg[i]; }      // (x * 1.0f == x)
        if (a.reduce_weight > 0.0f) a.g_cam[8] = a.reduce_weight;
    }
    __syncthreads();
    if (a.adam.p) {                         // (single GPU, one chunk: reduce_weight == 0, the message IS the gradient and the loss)
        const float t = t0 + 1.0f;
        if (threadIdx.x < 7) stepped[threadIdx.x] = adam_one(a.adam, threadIdx.x, t, m0, v0, p0, msg[threadIdx.x]);
        __syncthreads();
        if (threadIdx.x == 0) {
            a.adam.step[0] = t;
            if (a.adam.best && msg[7] < best0) {       // keep_best with the loss, the old minimum and the new camera at hand
                a.adam.best[0] = msg[7];
                for (int i = 0; i < 7; ++i) a.adam.best[1 + i] = stepped[i];
            }
        }
    }
}

}  // namespace nsa

extern "C" {

int nsa_cam_to_pose(const float* cam, uint32_t b, float* pose, nsa_stream_t stream) {
    using namespace nsa;
    if (b == 0) return NSA_OK;
    if (!cam || !pose) return NSA_EBADARG;
    launch_begin();
    hipLaunchKernelGGL(k_cam_to_pose, dim3((b + 63) / 64), dim3(64), 0, (hipStream_t)stream, cam, pose, b);
    return launch_end();
}

int nsa_pose_grad_to_cam(const float* cam, const float* g_pose, uint32_t b, float* g_cam, nsa_stream_t stream) {
    using namespace nsa;
    if (b == 0) return NSA_OK;
    if (!cam || !g_pose || !g_cam) return NSA_EBADARG;
    launch_begin();
    hipLaunchKernelGGL(k_pose_grad_to_cam, dim3((b + 63) / 64), dim3(64), 0, (hipStream_t)stream, cam, g_pose, g_cam, b);
    return launch_end();
}

int nsa_l1_loss(const float* pred, const float* target, uint32_t n, float* loss, float* g_pred, nsa_stream_t stream) {
    using namespace nsa;
    if (!pred || !target || !loss || !g_pred || n == 0) return NSA_EBADARG;
    launch_begin();
    hipLaunchKernelGGL(k_l1_loss, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, target, loss, g_pred, n);
    return launch_end();
}

int nsa_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* step, uint32_t n, float lr,
                  float beta1, float beta2, float eps, uint32_t lr_step, float lr_gamma, nsa_stream_t stream) {
    using namespace nsa;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step || n == 0 || n > 256) return NSA_EBADARG;
    AdamArgs a{param, grad, exp_avg, exp_avg_sq, step, nullptr, n, lr, beta1, beta2, eps, lr_gamma, lr_step, nullptr, nullptr, nullptr};
    launch_begin();
    hipLaunchKernelGGL(k_adam, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    return launch_end();
}

int nsa_adam_step_scaled(float* param, const float* grad, const float* grad_div, float* exp_avg, float* exp_avg_sq,
                         float* step, uint32_t n, float lr, float beta1, float beta2, float eps, uint32_t lr_step,
                         float lr_gamma, const float* loss, float* best, nsa_stream_t stream) {
    using namespace nsa;
    if (!param || !grad || !grad_div || !exp_avg || !exp_avg_sq || !step || n == 0 || n > 256) return NSA_EBADARG;
    if (best && !loss) return NSA_EBADARG;
    AdamArgs a{param, grad, exp_avg, exp_avg_sq, step, grad_div, n, lr, beta1, beta2, eps, lr_gamma, lr_step, best, loss, grad_div};
    launch_begin();
    hipLaunchKernelGGL(k_adam, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    return launch_end();
}

int nsa_track_head(const float* uv, const float* K, const float* cam, uint32_t n, float* pose, float* rays_o, float* rays_d,
                   float* depth_scale, nsa_stream_t stream) {
    using namespace nsa;
    if (!uv || !K || !cam || !pose || !rays_o || !rays_d || !depth_scale || n == 0) return NSA_EBADARG;
    TrackArgs a{};
    a.uv = uv; a.K = K; a.cam = const_cast<float*>(cam); a.pose = pose; a.n = n;
    a.rays_o = rays_o; a.rays_d = rays_d; a.depth_scale = depth_scale;
    launch_begin();
    hipLaunchKernelGGL(k_track_head, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return launch_end();
}

int nsa_track_tail(const float* uv, const float* K, float* cam, uint32_t n, const float* g_rays_o, const float* g_rays_d,
                   float* g_cam, int do_adam, float reduce_weight, float* exp_avg, float* exp_avg_sq, float* step, float lr,
                   float beta1, float beta2, float eps, uint32_t lr_step, float lr_gamma, const float* loss, float* best,
                   nsa_stream_t stream) {
    using namespace nsa;
    if (!uv || !K || !cam || !g_rays_o || !g_rays_d || !g_cam || n == 0) return NSA_EBADARG;
    if (do_adam && (!exp_avg || !exp_avg_sq || !step)) return NSA_EBADARG;
    if (best && (!loss || !do_adam)) return NSA_EBADARG;
    TrackArgs a{};
    a.uv = uv; a.K = K; a.cam = cam; a.n = n; a.g_o = g_rays_o; a.g_d = g_rays_d; a.g_cam = g_cam;
    a.reduce_weight = reduce_weight;
    if (do_adam) a.adam = AdamArgs{cam, g_cam, exp_avg, exp_avg_sq, step, nullptr, 7, lr, beta1, beta2, eps, lr_gamma, lr_step,
                                    best, loss, nullptr};
    launch_begin();
    hipLaunchKernelGGL(k_track_tail, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    return launch_end();
}

int nsa_track_begin(const float* uv_in, const float* gt_in, float* uv, float* gt, const float* K, const float* cam, uint32_t n,
                    float* pose, float* rays_o, float* rays_d, float* depth_scale, nsa_stream_t stream) {
    using namespace nsa;
    if (!uv_in || !gt_in || !uv || !gt || !K || !cam || !pose || !rays_o || !rays_d || !depth_scale || n == 0) return NSA_EBADARG;
    TrackArgs a{};
    a.K = K; a.cam = const_cast<float*>(cam); a.pose = pose; a.n = n;
    a.rays_o = rays_o; a.rays_d = rays_d; a.depth_scale = depth_scale;
    launch_begin();
    hipLaunchKernelGGL(k_track_begin, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, uv_in, gt_in, uv, gt);
    return launch_end();
}

int nsa_track_begin_draw(const float* uv_in, const float* gt_in, float* uv, float* gt, const float* K, const float* cam, uint32_t n,
                         float* pose, float* rays_o, float* rays_d, float* depth_scale, uint64_t* state, uint64_t n_rand, float* t_rand,
                         uint32_t E, uint32_t n_extra, uint32_t S, int32_t* extra_idx, nsa_stream_t stream) {
    using namespace nsa;
    if (!uv_in || !gt_in || !uv || !gt || !K || !cam || !pose || !rays_o || !rays_d || !depth_scale || n == 0) return NSA_EBADARG;
    if (!draw_args_ok(state, n_rand, t_rand, E, n_extra, n, S, extra_idx, nullptr)) return NSA_EBADARG;
    TrackArgs a{};
    a.K = K; a.cam = const_cast<float*>(cam); a.pose = pose; a.n = n;
    a.rays_o = rays_o; a.rays_d = rays_d; a.depth_scale = depth_scale;
    DrawArgs d{};
    const uint32_t draw_blocks = draw_launch_shape(d, reinterpret_cast<unsigned long long*>(state), n_rand, t_rand, E, n_extra, n, S,
                                                   extra_idx, nullptr);
    const uint32_t begin_blocks = (n + 255) / 256;
    launch_begin();
    hipLaunchKernelGGL(k_track_begin_draw, dim3(begin_blocks + draw_blocks), dim3(256), 0, (hipStream_t)stream, a, uv_in, gt_in, uv, gt,
                       d, begin_blocks);
    return launch_end();
}

uint64_t nsa_track_finish_workspace(uint32_t n) { return ((uint64_t)(n + 3) / 4) * 16 + 4; }

int nsa_track_finish(const float* uv, const float* K, float* cam, uint32_t n, uint32_t S, const float* z_vals, const float* g_x,
                     const float* g_dir, const float* ray_loss, float* g_cam, int do_adam, float reduce_weight, float* exp_avg,
                     float* exp_avg_sq, float* step, float lr, float beta1, float beta2, float eps, uint32_t lr_step,
                     float lr_gamma, float* best, float* workspace, nsa_stream_t stream) {
    using namespace nsa;
    if (!uv || !K || !cam || !z_vals || !g_x || !g_dir || !ray_loss || !g_cam || !workspace || n == 0 || S == 0) return NSA_EBADARG;
    if (do_adam && (!exp_avg || !exp_avg_sq || !step)) return NSA_EBADARG;
    if (best && !do_adam) return NSA_EBADARG;
    FinishArgs f{};
    f.t.uv = uv; f.t.K = K; f.t.cam = cam; f.t.n = n; f.t.g_cam = g_cam; f.t.reduce_weight = reduce_weight;
    if (do_adam) f.t.adam = AdamArgs{cam, g_cam, exp_avg, exp_avg_sq, step, nullptr, 7, lr, beta1, beta2, eps, lr_gamma, lr_step,
                                      best, g_cam + 7, nullptr};
    f.z_vals = z_vals; f.g_x = g_x; f.g_dir = g_dir; f.ray_loss = ray_loss; f.S = S;
    f.inv_n = 1.0f / (float)(3 * (uint64_t)n);
    const uint32_t blocks = (n + FIN_W - 1) / FIN_W;
    f.ticket = reinterpret_cast<unsigned*>(workspace);       // first 4 floats: the ticket (zero-filled by the caller once)
    f.part = workspace + 4;
    launch_begin();
    hipLaunchKernelGGL(k_track_finish, dim3(blocks), dim3(64 * FIN_W), 0, (hipStream_t)stream, f);
    return launch_end();
}

}  // extern "C"
