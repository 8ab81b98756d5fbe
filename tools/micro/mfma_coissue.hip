// Micro-benchmark: do the matrix pipe and the VALU overlap ACROSS waves of one SIMD?  A 512-thread workgroup places waves w and
// w + 4 on the same SIMD; waves 0..3 issue only MFMAs (32x32x16 bf16), waves 4..7 only independent v_fma_f32.  Times: both roles
// together vs each role alone (the other half of the workgroup exits at once).  Perfect overlap = max of the two, none = their sum.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_coissue.hip -o mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;

// mode bit 0: MFMA waves run, bit 1: VALU waves run;  PRIO: raise the priority of the VALU waves (s_setprio 1) or of the MFMA waves (2)
template <int KV, int PRIO>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool mfma_role = wave < 4;
    bf16x8_t a, b;
    {
        uint4 u = reinterpret_cast<const uint4*>(in)[threadIdx.x & 63];
        __builtin_memcpy(&a, &u, 16);
        u = reinterpret_cast<const uint4*>(in)[64 + (threadIdx.x & 63)];
        __builtin_memcpy(&b, &u, 16);
    }
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = in[128 + i + (threadIdx.x & 63)];
    const float c = in[200], d = in[201];
    if (PRIO == 1 && !mfma_role) __builtin_amdgcn_s_setprio(1);
    if (PRIO == 2 && mfma_role) __builtin_amdgcn_s_setprio(1);
    if (mfma_role) {
        if (mode & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < 16; ++m) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
            }
    } else {
        if (mode & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int v = 0; v < 16 * KV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(c), "v"(d));
            }
    }
    asm volatile("s_nop 15\n s_nop 15");
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][7];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KV, int PRIO>
void run(float* out, float* in) {
    const int iters = 2000;
    float ms[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<KV, PRIO>), dim3(256), dim3(512), 0, 0, out, in, 10, mode);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KV, PRIO>), dim3(256), dim3(512), 0, 0, out, in, iters, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[mode], e0, e1);
    }
    const double cyc = 2.4e9 * 1e-3 / (2000.0 * 16);
    printf("VALU per MFMA slot %2d  prio %d : MFMA alone %.3f ms (%.1f cyc/MFMA)  VALU alone %.3f ms (%.2f cyc/VALU)  together %.3f ms  "
           "-> overlap %.0f %% of the shorter\n", KV, PRIO, ms[1], ms[1] * cyc, ms[2], ms[2] * cyc / KV, ms[3],
           100.0 * (ms[1] + ms[2] - ms[3]) / (ms[1] < ms[2] ? ms[1] : ms[2]));
}

int main() {
    float *out, *in;
    hipMalloc(&out, 4 << 20); hipMalloc(&in, 1 << 16);
    hipMemset(in, 0, 1 << 16);
    run<4, 0>(out, in); run<7, 0>(out, in); run<12, 0>(out, in);
    run<7, 1>(out, in); run<7, 2>(out, in);
    return 0;
}
