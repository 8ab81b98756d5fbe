// Micro-benchmark for DESIGN 4.1 / VERDICT r3 #6: are packed-fp32 VALU instructions (v_pk_fma_f32, what the SLP vectoriser makes of
// adjacent fp32 math) reliable next to MFMAs on gfx950?  Every wave runs two recurrences on the same data,
//     y <- y * c + d      once with v_pk_fma_f32 on a register pair, once with two v_fma_f32,
// (IEEE fma either way: the results must be bit-identical) while matrix instructions are in flight
//   mode 0  no MFMA at all                                   mode 1  the SAME wave interleaves v_mfma_f32_16x16x32_bf16
//   mode 2  the same with v_mfma_f32_32x32x16_bf16           mode 3  role split: waves 0..3 of the workgroup issue only 16x16x32 MFMAs,
//                                                                    waves 4..7 (their SIMD partners) only the two recurrences
//   mode 4  as 3 with 32x32x16
// NOPS = s_nop states between consecutive instructions of the recurrences (0 or 3: the `-mllvm -amdgpu-snop-padding=2` spacing that
// made the SLP builds of the quad kernels irreproducible).  Output: lanes whose packed and scalar results differ, per mode.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/pk_mfma_interference.hip -o tools/micro/pk_mfma_interference.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;

template <int MODE, int NOPS>
__global__ __launch_bounds__(512) void k(const float* in, int iters, unsigned* bad_lanes, unsigned* bad_hist, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool split = MODE == 3 || MODE == 4;
    const bool mfma_role = !split || wave < 4;
    const bool pk_role = !split || wave >= 4;
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
    // recurrence data: contraction keeps the values bounded, different per lane and per half
    f32x2 y[4], c2, d2;
    float ys[8];
    const float cv = 0.9990234375f - 0.0001f * (lane & 7), dv = 0.001f * (1 + (lane & 3));
    c2 = {cv, cv * 0.5f};
    d2 = {dv, dv * 3.0f};
    for (int i = 0; i < 4; ++i) {
        y[i] = {1.0f + 0.01f * lane + i, 2.0f - 0.02f * lane + i};
        ys[2 * i] = y[i].x;
        ys[2 * i + 1] = y[i].y;
    }
#define PAD if (NOPS) asm volatile("s_nop 2");
    for (int it = 0; it < iters; ++it) {
        if (pk_role) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(c2), "v"(d2));
                PAD
            }
        }
        if (mfma_role) {
            if (MODE == 1 || MODE == 3) {
#pragma unroll
                for (int m = 0; m < 4; ++m) acc4[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[m], 0, 0, 0);
            } else if (MODE == 2 || MODE == 4) {
#pragma unroll
                for (int m = 0; m < 2; ++m) acc16[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc16[m], 0, 0, 0);
            }
        }
        if (pk_role) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ys[2 * i]) : "v"(c2.x), "v"(d2.x));
                PAD
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ys[2 * i + 1]) : "v"(c2.y), "v"(d2.y));
                PAD
            }
        }
    }
    asm volatile("s_nop 15\n s_nop 15");
    bool bad = false;
    if (pk_role)
        for (int i = 0; i < 4; ++i)
            bad |= __float_as_uint(y[i].x) != __float_as_uint(ys[2 * i]) || __float_as_uint(y[i].y) != __float_as_uint(ys[2 * i + 1]);
    if (bad) {
        atomicAdd(bad_lanes, 1u);
        atomicAdd(&bad_hist[lane >> 4], 1u);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) s += acc4[j][0] + acc4[j][3];
    for (int j = 0; j < 2; ++j) s += acc16[j][0] + acc16[j][9];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s + y[0].x + ys[0];
}

template <int MODE, int NOPS>
void run(const float* in, unsigned* cnt, float* sink, const char* what) {
    hipMemset(cnt, 0, 64);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<MODE, NOPS>), dim3(1024), dim3(512), 0, 0, in, 4000, cnt, cnt + 1, sink);
    hipDeviceSynchronize();
    unsigned h[8];
    hipMemcpy(h, cnt, 32, hipMemcpyDeviceToHost);
    printf("mode %d (%s), s_nop padding %s: %u lanes of %u with packed != scalar   [by lane quarter 0-15/16-31/32-47/48-63: %u %u %u %u]\n",
           MODE, what, NOPS ? "3 states" : "none", h[0], 5u * 1024u * 512u, h[1], h[2], h[3], h[4]);
}

int main() {
    float *in, *sink;
    unsigned* cnt;
    hipMalloc(&in, 1 << 16);
    hipMalloc(&sink, 1024 * 512 * 4);
    hipMalloc(&cnt, 64);
    hipMemset(in, 0x3c, 1 << 16);      // bf16 0x3c3c ~ 0.0115: finite products
    run<0, 0>(in, cnt, sink, "no MFMA");
    run<1, 0>(in, cnt, sink, "same wave, 16x16x32");
    run<1, 1>(in, cnt, sink, "same wave, 16x16x32");
    run<2, 0>(in, cnt, sink, "same wave, 32x32x16");
    run<2, 1>(in, cnt, sink, "same wave, 32x32x16");
    run<3, 0>(in, cnt, sink, "SIMD partner issues 16x16x32");
    run<3, 1>(in, cnt, sink, "SIMD partner issues 16x16x32");
    run<4, 0>(in, cnt, sink, "SIMD partner issues 32x32x16");
    run<4, 1>(in, cnt, sink, "SIMD partner issues 32x32x16");
    return 0;
}
