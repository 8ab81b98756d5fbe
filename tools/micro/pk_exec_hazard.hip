// Second micro-benchmark for DESIGN 4.2 (after pk_mfma_interference.hip found the packed instruction itself reliable beside MFMAs):
// the CONTEXT in which the SLP build of the quad kernels uses v_pk_*_f32 -- the trilinear corner blend of the fine grid:
//     if (inside) { eight gathered rows -> weights x rows (packed: two channels per instruction) -> sum }      (per-lane exec mask)
// with dependent-accumulator MFMAs of the same wave and of its SIMD partner in flight.  Every lane computes the blend twice from the
// same loaded values -- with float2 vector arithmetic (hipcc emits v_pk_mul_f32 / v_pk_fma_f32 and its own hazard padding, exactly
// as in the product scenario) and with scalar fmaf on the components -- and the two must agree bit for bit.
//   MASK   0: all lanes inside   1: a lane-varying, iteration-varying predicate (divergent region, as `inside`)
//   LOADS  0: operands from registers   1: operands gathered from a 64 MiB table inside the region (s_waitcnt right before the packed op)
//   MFMA   0: none   1: v_mfma_f32_16x16x32_bf16 chains (4 accumulators) around the region   2: 32x32x16
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/micro/pk_exec_hazard.hip -o tools/micro/pk_exec_hazard.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;

__device__ __forceinline__ uint32_t mix(uint32_t a) {
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}

template <int MASK, int LOADS, int MFMA>
__global__ __launch_bounds__(512) void k(const float* __restrict__ table, uint32_t rows_mask, const float* in, int iters,
                                         unsigned* bad, float* sink) {
    const int lane = threadIdx.x & 63;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8_t a, b;
    {
        uint4 u = reinterpret_cast<const uint4*>(in)[lane];
        __builtin_memcpy(&a, &u, 16);
        u = reinterpret_cast<const uint4*>(in)[64 + lane];
        __builtin_memcpy(&b, &u, 16);
    }
    f32x4 acc4[4];
    f32x16 acc16[2];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc4[j][r] = 0.f;
    for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc16[j][r] = 0.f;
    f32x2 sum_v = {0.f, 0.f};
    float sum_s0 = 0.f, sum_s1 = 0.f;
    unsigned mism = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t hsh = mix(tid * 2654435761u + it);
        const float t0 = (float)(hsh & 1023) * (1.0f / 1024.0f), t1 = (float)((hsh >> 10) & 1023) * (1.0f / 1024.0f),
                    t2 = (float)((hsh >> 20) & 1023) * (1.0f / 1024.0f);
        if (MFMA == 1) {
#pragma unroll
            for (int m = 0; m < 4; ++m) acc4[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[m], 0, 0, 0);
        } else if (MFMA == 2) {
#pragma unroll
            for (int m = 0; m < 2; ++m) acc16[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc16[m], 0, 0, 0);
        }
        const bool inside = MASK == 0 || ((hsh >> 7) & 3) != 0;            // three quarters of the lanes, different ones every iteration
        f32x2 out_v = {0.f, 0.f};
        float out_s0 = 0.f, out_s1 = 0.f;
        if (inside) {
            const float w[3] = {t0 * t0 * (3.f - 2.f * t0), t1 * t1 * (3.f - 2.f * t1), t2 * t2 * (3.f - 2.f * t2)};
            f32x2 v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (LOADS) {
                    const uint32_t row = (mix(hsh + c * 0x9E3779B9u)) & rows_mask;
                    v[c] = *reinterpret_cast<const f32x2*>(table + (size_t)row * 2);
                } else {
                    v[c] = {t0 + (float)c, t1 - (float)c};
                }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float wt = ((c & 1) ? w[0] : 1.f - w[0]) * ((c & 2) ? w[1] : 1.f - w[1]) * ((c & 4) ? w[2] : 1.f - w[2]);
                const f32x2 w2 = {wt, wt};
                out_v = __builtin_elementwise_fma(w2, v[c], out_v);        // v_pk_fma_f32 (broadcast weight: op_sel_hi form)
                out_s0 = fmaf(wt, v[c].x, out_s0);
                out_s1 = fmaf(wt, v[c].y, out_s1);
            }
        }
        if (__float_as_uint(out_v.x) != __float_as_uint(out_s0) || __float_as_uint(out_v.y) != __float_as_uint(out_s1)) ++mism;
        sum_v += out_v;
        sum_s0 += out_s0;
        sum_s1 += out_s1;
    }
    if (mism) atomicAdd(bad, mism);
    float s = sum_v.x + sum_v.y + sum_s0 + sum_s1;
    for (int j = 0; j < 4; ++j) s += acc4[j][0];
    for (int j = 0; j < 2; ++j) s += acc16[j][0];
    sink[tid] = s;
}

template <int MASK, int LOADS, int MFMA>
void run(const float* table, uint32_t rows_mask, const float* in, unsigned* cnt, float* sink) {
    (void)hipMemset(cnt, 0, 16);
    const int iters = 400;
    for (int rep = 0; rep < 5; ++rep)
        hipLaunchKernelGGL((k<MASK, LOADS, MFMA>), dim3(1024), dim3(512), 0, 0, table, rows_mask, in, iters, cnt, sink);
    (void)hipDeviceSynchronize();
    unsigned h = 0;
    (void)hipMemcpy(&h, cnt, 4, hipMemcpyDeviceToHost);
    printf("exec mask %-9s operands %-8s MFMA %-9s : %u blends of %llu with packed != scalar\n", MASK ? "divergent" : "full",
           LOADS ? "gathered" : "register", MFMA == 0 ? "none" : MFMA == 1 ? "16x16x32" : "32x32x16", h,
           5ull * 1024 * 512 * iters);
}

int main() {
    float *table, *in, *sink;
    unsigned* cnt;
    const uint32_t rows = 1u << 23;                       // 8 M rows x 8 B = 64 MiB
    (void)hipMalloc(&table, (size_t)rows * 8);
    (void)hipMalloc(&in, 1 << 16);
    (void)hipMalloc(&sink, 1024 * 512 * 4);
    (void)hipMalloc(&cnt, 64);
    (void)hipMemset(in, 0x3c, 1 << 16);
    {   // table values: small finite floats
        float* h = (float*)malloc((size_t)rows * 8);
        uint32_t s = 12345u;
        for (size_t i = 0; i < (size_t)rows * 2; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 9) - (1 << 22)) * (1.0f / (1 << 22)) * 0.05f; }
        (void)hipMemcpy(table, h, (size_t)rows * 8, hipMemcpyHostToDevice);
        free(h);
    }
    run<0, 0, 0>(table, rows - 1, in, cnt, sink);
    run<1, 0, 0>(table, rows - 1, in, cnt, sink);
    run<0, 1, 0>(table, rows - 1, in, cnt, sink);
    run<1, 1, 0>(table, rows - 1, in, cnt, sink);
    run<1, 1, 1>(table, rows - 1, in, cnt, sink);
    run<1, 1, 2>(table, rows - 1, in, cnt, sink);
    run<0, 1, 1>(table, rows - 1, in, cnt, sink);
    run<1, 0, 1>(table, rows - 1, in, cnt, sink);
    return 0;
}
