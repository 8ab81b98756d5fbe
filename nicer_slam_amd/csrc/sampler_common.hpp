// sampler_common.hpp -- arguments and per-point setup shared by the sampler's SDF-pass kernels (render_sampler.hip: every wave
// runs the whole per-point program; render_sampler_ws.hip: wave-specialised producer / consumer form).
// Reference: UniformSampler.get_z_vals / near_far_from_cube (code/model/ray_sampler.py:23-61).
#pragma once
#include "sdf_net.hpp"

namespace nsa {

struct SamplerArgs {
    const float* rays_o;      // [R,3]
    const float* rays_d;      // [R,3]
    const float* t_lin;       // [E] = linspace(0,1,E)
    const float* t_rand;      // [R,E] stratified jitter in [0,1) or nullptr (eval mode)
    float* z;                 // [R,E] out
    float* sdf;               // [R,E] out
    float* far;               // [R] out
    uint32_t R, E;
    float near, bound, far_cap;
    const float* table_c;
    const float* table_f;
    const float* wp_c;
    const float* wp_f;
    float df_c, df_f;
};

// far end of the ray inside the cube [-bound, bound]^3, clamped to far_cap (ray_sampler.py:23-35).
__device__ __forceinline__ float cube_far(const float (&o)[3], const float (&d)[3], float bound, float far_cap, float near_clamp) {
    float nearv = -INFINITY, farv = INFINITY;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float den = d[k] + 1e-15f;
        const float t0 = (-bound - o[k]) / den;
        const float t1 = (bound - o[k]) / den;
        nearv = fmaxf(nearv, t0 < t1 ? t0 : t1);
        farv = fminf(farv, t0 > t1 ? t0 : t1);
    }
    if (farv < nearv) farv = 1e9f;
    (void)near_clamp;
    return fminf(farv, far_cap);
}

#ifndef NSA_OCC_SAMPLER
#define NSA_OCC_SAMPLER 2      // (asks for <= 256 registers; the kernel needs 167: three waves per SIMD.  4 = 128 registers spills 60+)
#endif

// ray index of point `pid` (pid < R * E): a 32-bit division whenever the point count allows it (always at the shipped sizes; the
// 64-bit one is ~100 vector instructions per tile)
__device__ __forceinline__ uint32_t ray_of_point(uint64_t pid, uint32_t E, uint64_t total) {
    return total <= 0xFFFFFFFFull ? (uint32_t)pid / E : (uint32_t)(pid / E);
}

// sample position i of ray `ray`: stratified z and the point (ray_sampler.py:49-59); every product rounded separately, as the
// reference's elementwise torch ops do (see mul_rn)
struct RayOfTile {
    float o[3], d[3], farv;
};
__device__ __forceinline__ void ray_of_tile(const SamplerArgs& a, uint32_t ray, RayOfTile& r) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.o[k] = a.rays_o[ray * 3 + k]; r.d[k] = a.rays_d[ray * 3 + k]; }
    r.farv = cube_far(r.o, r.d, a.bound, a.far_cap, a.near);
}
__device__ __forceinline__ void sampler_point(const SamplerArgs& a, uint64_t pid, const RayOfTile& r, uint32_t i, float (&x)[3],
                                              float& zi, float& farv) {
    const float (&o)[3] = r.o;
    const float (&d)[3] = r.d;
    farv = r.farv;
    const float nearv = a.near;
    // z_lin(i) = near (1 - t_i) + far t_i ; stratified: lower + (upper - lower) * rand
    const uint32_t E = a.E;
    const float ti = a.t_lin[i];
    zi = mul_rn(nearv, 1.0f - ti) + mul_rn(farv, ti);
    if (a.t_rand) {
        const float tp = a.t_lin[i + 1 < E ? i + 1 : i], tm = a.t_lin[i > 0 ? i - 1 : 0];
        const float zp = mul_rn(nearv, 1.0f - tp) + mul_rn(farv, tp);
        const float zm = mul_rn(nearv, 1.0f - tm) + mul_rn(farv, tm);
        const float upper = i + 1 < E ? 0.5f * (zp + zi) : zi;
        const float lower = i > 0 ? 0.5f * (zi + zm) : zi;
        zi = lower + mul_rn(upper - lower, a.t_rand[pid]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = o[k] + mul_rn(zi, d[k]);
}

}  // namespace nsa
