// Micro-benchmark: the random-row gather ceiling of MI355X for the colour grid's hashed levels (8-byte rows in 128-MiB levels).
// Every lane reads N independent rows at hashed indices (no reuse), U loads in flight per lane, all CUs busy; reports rows/s,
// and the same with each lane reading the x-neighbour PAIR of an even cell (two adjacent 8-byte rows = one 16-byte load), which
// is what a dense level / an even hashed cell gives.  Table sizes: one level (128 MiB), the six hashed levels (768 MiB).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gather_bench.hip -o tools/micro/gather_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint32_t rng(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int U, int BYTES>
__global__ __launch_bounds__(256) void k(const float* __restrict__ t, uint32_t rows, uint32_t n_per_lane, float* out) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.0f;
    for (uint32_t i = 0; i < n_per_lane; i += U) {
        float v[U][BYTES / 4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t row = rng(gid * n_per_lane + i + u) % rows;
            if (BYTES == 16) row &= ~1u;
            const float* p = t + (size_t)row * 2;
            if (BYTES == 8) { const float2 x = *reinterpret_cast<const float2*>(p); v[u][0] = x.x; v[u][1] = x.y; }
            else { const float4 x = *reinterpret_cast<const float4*>(p); v[u][0] = x.x; v[u][1] = x.y; v[u][2] = x.z; v[u][3] = x.w; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < BYTES / 4; ++c) acc += v[u][c];
    }
    if (acc == 123.456f) out[gid] = acc;
}

template <int U, int BYTES>
void run(const float* t, uint32_t rows, float* out, const char* what) {
    const uint32_t blocks = 256 * 12, tpb = 256, npl = 64;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<U, BYTES>), dim3(blocks), dim3(tpb), 0, 0, t, rows, npl, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<U, BYTES>), dim3(blocks), dim3(tpb), 0, 0, t, rows, npl, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
    const double loads = (double)blocks * tpb * npl;
    printf("%-34s table %4u MiB  U=%2d  %2d-byte loads: %7.3f ms  %6.1f G loads/s  %6.1f G rows/s  (%5.2f TB/s of 64-B lines if none shared)\n",
           what, (unsigned)((uint64_t)rows * 8 >> 20), U, BYTES, ms, loads / ms * 1e-6, loads * (BYTES / 8) / ms * 1e-6,
           loads * 64 / ms * 1e-9);
}

int main() {
    float* t; float* out;
    const uint32_t rows_big = 96u << 20;           // 768 MiB of 8-byte rows
    hipMalloc(&t, (size_t)rows_big * 8); hipMalloc(&out, 256 * 12 * 256 * 4);
    hipMemset(t, 0, (size_t)rows_big * 8);
    for (uint32_t rows : {16u << 20, rows_big}) {
        run<4, 8>(t, rows, out, "random 8-byte rows");
        run<8, 8>(t, rows, out, "random 8-byte rows");
        run<16, 8>(t, rows, out, "random 8-byte rows");
        run<8, 16>(t, rows, out, "random even-x row pairs");
        run<16, 16>(t, rows, out, "random even-x row pairs");
    }
    return 0;
}
