// Micro-benchmark: float atomic scatter of C-channel rows, lane-per-row vs lane-per-channel issue patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ inline uint32_t rng(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int C, bool TRANSPOSED>
__global__ void k(float* t, uint32_t rows, uint32_t n_per_lane, uint32_t locality) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    for (uint32_t i = 0; i < n_per_lane; ++i) {
        uint32_t key = gid * n_per_lane + i;
        if (locality) key = (key / locality);          // `locality` consecutive lanes share a row -> contention
        const uint32_t row = rng(key) % rows;
        if (!TRANSPOSED) {
#pragma unroll
            for (int c = 0; c < C; ++c) atomicAdd(t + (size_t)row * C + c, 1.0f);
        } else {
            // round j: lane group g = lane / C handles the row of lane g*C + j; lane % C = channel
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const uint32_t r = __shfl(row, (lane / C) * C + j);
                atomicAdd(t + (size_t)r * C + (lane % C), 1.0f);
            }
        }
    }
}

// lane-per-channel, but 2C adjacent lanes cover the two ADJACENT rows (row, row+1) = 2C contiguous floats
template <int C>
__global__ void kpair(float* t, uint32_t rows, uint32_t n_per_lane) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    for (uint32_t i = 0; i < n_per_lane; ++i) {
        const uint32_t row = (rng(gid * n_per_lane + i) % (rows - 1)) & ~1u;    // even row: the pair shares a 2C-float span
#pragma unroll
        for (int j = 0; j < 2 * C; ++j) {     // each lane's row pair is sent in round j by its 2C-lane group
            const uint32_t r = __shfl(row, (lane / (2 * C)) * (2 * C) + j);
            atomicAdd(t + (size_t)r * C + (lane % (2 * C)), 1.0f);
        }
    }
}

template <int C>
void run_pair(float* t, uint32_t rows) {
    const uint32_t blocks = 4096, tpb = 256, npl = 16;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((kpair<C>), dim3(blocks), dim3(tpb), 0, 0, t, rows, npl);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((kpair<C>), dim3(blocks), dim3(tpb), 0, 0, t, rows, npl);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double nrows = (double)blocks * tpb * npl * 2;          // every lane sends a PAIR of rows
    printf("%-40s C=%d rows=%9u         %8.3f ms  %7.2f G rows/s (adjacent row pairs)\n", "lane-per-channel, row pairs", C, rows, ms,
           nrows / ms * 1e-6);
}

template <int C, bool T>
void run(const char* name, float* t, uint32_t rows, uint32_t locality) {
    const uint32_t blocks = 4096, tpb = 256, npl = 16;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<C, T>), dim3(blocks), dim3(tpb), 0, 0, t, rows, npl, locality);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<C, T>), dim3(blocks), dim3(tpb), 0, 0, t, rows, npl, locality);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double n = (double)blocks * tpb * npl * C;
    printf("%-40s C=%d rows=%9u loc=%2u  %8.3f ms  %7.2f G float-atomics/s  %7.2f G rows/s\n", name, C, rows, locality, ms,
           n / ms * 1e-6, n / C / ms * 1e-6);
}

int main() {
    float* t; const size_t bytes = (size_t)1 << 30;
    hipMalloc(&t, bytes); hipMemset(t, 0, bytes);
    for (uint32_t loc : {0u, 8u}) {
        run<8, false>("lane-per-row", t, 36000, loc);
        run<8, true>("lane-per-channel", t, 36000, loc);
        run<8, false>("lane-per-row", t, 1u << 22, loc);
        run<8, true>("lane-per-channel", t, 1u << 22, loc);
        run<4, false>("lane-per-row", t, 1u << 20, loc);
        run<4, true>("lane-per-channel", t, 1u << 20, loc);
        run<2, false>("lane-per-row", t, 1u << 27, loc);
        run<2, true>("lane-per-channel", t, 1u << 27, loc);
    }
    run_pair<8>(t, 36000); run_pair<4>(t, 1u << 20); run_pair<2>(t, 1u << 27);
    // verify sum
    std::vector<float> h(36000 * 8);
    hipMemset(t, 0, bytes);
    hipLaunchKernelGGL((k<8, true>), dim3(4096), dim3(256), 0, 0, t, 36000u, 16u, 0u);
    hipMemcpy(h.data(), t, h.size() * 4, hipMemcpyDeviceToHost);
    double s = 0; for (float v : h) s += v;
    printf("sum %.0f expect %.0f\n", s, 4096.0 * 256 * 16 * 8);
    return 0;
}
