// Micro-benchmark: issue cost of bf16 MFMAs with K independent VALU instructions between consecutive MFMAs, for the two bf16
// shapes (32x32x16: 16 K MAC / instruction, 16x16x32: 8 K MAC) at 1, 2 and 4 waves per SIMD.  Answers: how many VALU
// instructions does one MFMA hide, and what does an MFMA cost in VALU issue slots?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;

template <int SHAPE, int K, int NM>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
    bf16x8_t a, b;
    {
        uint4 u = reinterpret_cast<const uint4*>(in)[threadIdx.x & 63];
        __builtin_memcpy(&a, &u, 16);
        u = reinterpret_cast<const uint4*>(in)[64 + (threadIdx.x & 63)];
        __builtin_memcpy(&b, &u, 16);
    }
    f32x16 acc32[4];
    f32x4 acc16[8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[j][r] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[j][r] = 0.f;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = in[128 + i + threadIdx.x];
    const float c = in[200], d = in[201];
    // inline asm pins the program order: MFMA, then K independent v_fma_f32, repeated (the compiler would otherwise cluster
    // the MFMAs and pack the fmas)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (SHAPE == 32) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc32[m & 3]) : "v"(a), "v"(b));
            else if (SHAPE == 16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc16[m & 7]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < K; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(m * K + v) & 7]) : "v"(c), "v"(d));
        }
    }
    asm volatile("s_nop 15\n s_nop 15");
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc32[j][0] + acc32[j][7];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc16[j][1];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int K>
void run(float* out, float* in, int waves_per_simd) {
    constexpr int NM = 16;
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;            // 256 CUs x (4 waves per block = 1 per SIMD)
    hipLaunchKernelGGL((k<SHAPE, K, NM>), dim3(blocks), dim3(256), 0, 0, out, in, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, K, NM>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // cycles per (MFMA + K VALU) segment per SIMD, assuming 2.4 GHz and waves_per_simd resident waves per SIMD
    const double seg = (double)iters * NM * waves_per_simd;
    printf("shape %2d  K %d  waves/SIMD %d : %.3f ms  -> %.1f cycles per segment per SIMD (@2.4GHz)\n", SHAPE, K, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / seg);
}

int main() {
    float *out, *in;
    hipMalloc(&out, 4 << 20); hipMalloc(&in, 1 << 16);
    hipMemset(in, 0, 1 << 16);
    for (int w : {1, 2, 4}) {
        run<32, 0>(out, in, w); run<32, 2>(out, in, w); run<32, 4>(out, in, w); run<32, 5>(out, in, w); run<32, 6>(out, in, w);
        run<32, 8>(out, in, w); run<32, 12>(out, in, w);
        run<16, 0>(out, in, w); run<16, 1>(out, in, w); run<16, 2>(out, in, w); run<16, 3>(out, in, w); run<16, 4>(out, in, w);
        run<16, 6>(out, in, w);
        run<0, 4>(out, in, w); run<0, 8>(out, in, w);     // VALU only: 4 / 8 fmas per segment
    }
    return 0;
}
