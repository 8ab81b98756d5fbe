// emit_gemm.hip -- MLP weight gradients of a mapping iteration from the emission rows of the MAP backward kernels
// (C ABI section 4: nsa_emit_gemm).
//
// The MAP kernels write, per point, the vectors whose outer products are the weight gradients, as rows of a [rows][ld] fp32
// buffer (column = point; fused/mapping.py, structs SE<NH> / SE4<NH> / CE).  A weight gradient is then
//     out[m][n] = sum_p emit[a0 + m][p] * emit[b0 + n][p]   (+ a second product pair for the layers fed by two row sets)
// i.e. C = A * B^T with BOTH operands K-contiguous -- exactly the fragment order of v_mfma_f32_32x32x16_bf16 (a lane holds 8
// consecutive k of one row), so fragments come straight from global memory with 32-byte loads, no transposes, no LDS.  The
// bias gradients are the row sums of the A rows: one extra output column.
// Replaces torch.bmm over K-chunks (+ a sum over chunks) and the ATen row reductions of round 1 (reference: the autograd of
// base_networks.py:195-221,333-395 w.r.t. the Linear weights).
//
// Mapping: the reduction dimension (points) is cut into ~1024 slices of `kw` columns; ONE WAVE per (slice, group of up to three
// 32-column output tiles) computes all M <= 64 rows of its tiles, so every A fragment is split once per wave and serves up to
// three tiles (the first version gave each 32-column tile its own wave: five redundant splits of A and 2.6 TB/s).  A lane owns
// 8 consecutive points of one row per k-step (32-byte loads, register double buffer one k-step ahead); at one wave per SIMD the
// loads of 1024 waves in flight keep HBM busy.  Operands are fp32; products are fp32-faithful (three exact bf16 pieces, six
// MFMAs -- mlp_common.hpp).  The matrix cores add into `acc` over 256 columns only (long chains of dependent MFMA accumulations on
// one accumulator showed errors of ~3e-6 of sum |a b|); `total` collects those sub-sums in round-to-nearest fp32.  The bias
// gradients (row sums of A) are VALU sums of the A fragments the lane already holds.  Every wave writes its partial tile set;
// k_emit_reduce adds the partials in slice order, so the result is deterministic (no atomics).  Bound: reading the emission rows
// once from HBM, (M + N) * ld * 4 bytes per product pair.
#include "mlp_common.hpp"
#include "grid_common.hpp"

namespace nsa {

constexpr uint32_t KSUB = 256;        // columns per MFMA accumulation chain; ld is a multiple of 4096 (fused/mapping.py::emit_ld)
constexpr uint32_t SLICES = 1024;     // target number of K-slices (one wave each per tile group: one round at 1 wave/SIMD)
constexpr int NTW = 3;                // output tiles (of 32 columns) per wave

struct EmitGemmArgs {
    const float* emit;
    uint64_t ld;
    uint32_t a[2], b[2];              // first rows of the A / B operands of the (up to two) product pairs
    uint32_t pairs, M, N, sums;       // out is [M][N + sums]; column N = row sums of A (pair 0)
    uint32_t kw;                      // columns per slice (multiple of KSUB)
    float* partial;                   // [slices][M][N + sums]
    uint32_t slices, groups;          // grid = slices x groups workgroups, numbered by emit_block()
};

// Workgroup id -> (slice, column group).  The waves of one slice share its A rows; workgroups are dealt round-robin to the 8 XCDs, each
// with its own L2, so the groups of a slice are numbered 8 apart (same XCD) and back to back in dispatch order: the second group's
// A fragments are L2 hits instead of a second trip to HBM.  (With the former (slices, groups) grid they ran on the same XCD too, but a
// whole sweep of 1024 slices apart.)
__device__ __forceinline__ void emit_block(const EmitGemmArgs& g, uint32_t& slice, uint32_t& grp) {
    const uint32_t id = blockIdx.x;
    if ((g.slices & 7u) == 0) {
        const uint32_t per = 8u * g.groups;
        slice = (id / per) * 8u + (id & 7u);
        grp = (id >> 3) % g.groups;
    } else {
        slice = id / g.groups;
        grp = id % g.groups;
    }
}

__host__ __device__ inline uint32_t emit_kw(uint64_t ld) {
    const uint64_t n = ld / KSUB;
    return (uint32_t)((n + SLICES - 1) / SLICES) * KSUB;
}

__device__ __forceinline__ void load8(const float* __restrict__ p, float (&x)[8]) {
    const float4 u = reinterpret_cast<const float4*>(p)[0];
    const float4 v = reinterpret_cast<const float4*>(p)[1];
    x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = v.x; x[5] = v.y; x[6] = v.z; x[7] = v.w;
}

template <int MT, int NT>
struct EmitFrags {
    float a[MT][8], b[NT][8];
};

template <int MT, int NT>
__device__ __forceinline__ void emit_load(EmitFrags<MT, NT>& f, const float* const (&arow)[MT], const float* const (&brow)[NT],
                                          uint64_t k) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) load8(arow[mt] + k, f.a[mt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) load8(brow[nt] + k, f.b[nt]);
}

template <int MT, int NT>
__device__ __forceinline__ void emit_step(const EmitFrags<MT, NT>& f, f32x16 (&acc)[MT][NT], float (&asum)[MT]) {
    bf16x8_t ah[MT], am[MT], al[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        BFrag fa;
        split8(f.a[mt], fa);
        ah[mt] = as_bf16x8(fa.p[0]); am[mt] = as_bf16x8(fa.p[1]); al[mt] = as_bf16x8(fa.p[2]);
        asum[mt] += ((f.a[mt][0] + f.a[mt][1]) + (f.a[mt][2] + f.a[mt][3])) + ((f.a[mt][4] + f.a[mt][5]) + (f.a[mt][6] + f.a[mt][7]));
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        BFrag fb;
        split8(f.b[nt], fb);
        const bf16x8_t bh = as_bf16x8(fb.p[0]), bm = as_bf16x8(fb.p[1]), bl = as_bf16x8(fb.p[2]);
        // smallest terms first; the MT accumulators alternate so no MFMA waits on its predecessor
#define NSA_EMM(A, B)                                                                                  \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                              \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt], B, acc[mt][nt], 0, 0, 0);
        NSA_EMM(al, bh) NSA_EMM(ah, bl) NSA_EMM(am, bm) NSA_EMM(am, bh) NSA_EMM(ah, bm) NSA_EMM(ah, bh)
#undef NSA_EMM
    }
}

// grid: slices x (groups of NT output tiles) workgroups in emit_block() order; one wave per block
template <int MT, int NT>
__global__ __launch_bounds__(64) void k_emit_gemm(EmitGemmArgs g) {
    const int lane = threadIdx.x, i = lane & 31, kq = lane >> 5;
    const uint32_t N1 = g.N + g.sums;
    uint32_t slice, grp;
    emit_block(g, slice, grp);
    const uint32_t n0 = grp * (32u * NTW);                     // first output column of this wave
    const uint64_t c0 = (uint64_t)slice * g.kw;
    const uint64_t c1 = c0 + g.kw < g.ld ? c0 + g.kw : g.ld;
    f32x16 total[MT][NT];
    float asum[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        asum[mt] = 0.0f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) total[mt][nt][r] = 0.0f;
    }
    for (uint32_t pr = 0; pr < g.pairs; ++pr) {
        // rows past M / N are clamped to a valid row: their results are dropped at the store (every output column has its own
        // accumulator lanes, so nothing leaks)
        const float* arow[MT];
        const float* brow[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const uint32_t m = 32u * mt + i;
            arow[mt] = g.emit + (uint64_t)(g.a[pr] + (m < g.M ? m : g.M - 1)) * g.ld + 8u * kq;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const uint32_t n = n0 + 32u * nt + i;
            brow[nt] = g.emit + (uint64_t)(g.N ? g.b[pr] + (n < g.N ? n : g.N - 1) : g.a[pr]) * g.ld + 8u * kq;
        }
        float keep[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) keep[mt] = asum[mt];
        EmitFrags<MT, NT> f0, f1;
        emit_load<MT, NT>(f0, arow, brow, c0);
        for (uint64_t kb = c0; kb < c1; kb += KSUB) {
            f32x16 acc[MT][NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
            for (uint64_t k = kb; k < kb + KSUB; k += 32) {
                emit_load<MT, NT>(f1, arow, brow, k + 16);
                emit_step<MT, NT>(f0, acc, asum);
                if (k + 32 < c1) emit_load<MT, NT>(f0, arow, brow, k + 32);
                emit_step<MT, NT>(f1, acc, asum);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) total[mt][nt][r] += acc[mt][nt][r];
        }
        if (pr > 0) {          // the bias gradient is the row sum of the FIRST A block only
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asum[mt] = keep[mt];
        }
    }
    // D[row][col]: col = lane & 31 (the B row n), row = 8 (r >> 2) + 4 (lane >> 5) + (r & 3) (the A row within the m-tile)
    float* out = g.partial + (uint64_t)slice * g.M * N1;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const uint32_t n = n0 + 32u * nt + i;
        if (n < g.N) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t m = 32u * mt + 8u * (r >> 2) + 4u * kq + (r & 3);
                    if (m < g.M) out[(uint64_t)m * N1 + n] = total[mt][nt][r];
                }
        }
    }
    if (g.sums && grp == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float s = asum[mt] + __shfl_xor(asum[mt], 32);
            const uint32_t m = 32u * mt + i;
            if (kq == 0 && m < g.M) out[(uint64_t)m * N1 + g.N] = s;
        }
    }
}

// out[j] = sum over slices of partial[slice][j], in a fixed order: eight interleaved slice groups per output, combined in LDS
__global__ __launch_bounds__(256) void k_emit_reduce(const float* __restrict__ partial, float* __restrict__ out, uint32_t count,
                                                     uint32_t slices) {
    __shared__ float part[8][32];
    const uint32_t c = threadIdx.x & 31, q = threadIdx.x >> 5;
    const uint32_t j = blockIdx.x * 32u + c;
    // slice group q adds slices q, q + 8, q + 16, .. in eight interleaved chains: eight loads in flight per thread (the loop is bound
    // by load latency -- with two chains a reduction took 17 us for 25 MB of partials), summation order fixed
    float s[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) s[t] = 0.0f;
    if (j < count) {
        uint32_t sl = q;
        for (; sl + 56 < slices; sl += 64) {
#pragma unroll
            for (int t = 0; t < 8; ++t) s[t] += partial[(uint64_t)(sl + 8 * t) * count + j];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (sl + 8 * t < slices) s[t] += partial[(uint64_t)(sl + 8 * t) * count + j];
    }
    part[q][c] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (q == 0 && j < count) {
        float s = 0.0f;
#pragma unroll
        for (int t = 0; t < 8; ++t) s += part[t][c];
        out[j] = s;
    }
}

template <int MT>
static void emit_launch(EmitGemmArgs g, uint32_t slices, uint32_t tiles, hipStream_t st) {
    const uint32_t groups = tiles ? (tiles + NTW - 1) / NTW : 1;
    g.slices = slices; g.groups = groups;
    const uint32_t last = tiles ? tiles - (groups - 1) * NTW : 0;        // tiles of the last group
    // all groups but the last are full; instantiate by the widest group present (narrower groups clamp their rows)
    const uint32_t nt = groups > 1 ? NTW : (last ? last : 1);
    const dim3 grid(slices * groups), block(64);
    if (nt == 1) hipLaunchKernelGGL((k_emit_gemm<MT, 1>), grid, block, 0, st, g);
    else if (nt == 2) hipLaunchKernelGGL((k_emit_gemm<MT, 2>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((k_emit_gemm<MT, 3>), grid, block, 0, st, g);
}

}  // namespace nsa

extern "C" int nsa_emit_gemm(const float* emit, uint64_t ld, uint32_t pairs, const uint32_t* a_rows, const uint32_t* b_rows,
                             uint32_t M, uint32_t N, int row_sums, float* out, float* workspace, nsa_stream_t stream) {
    using namespace nsa;
    if (!emit || !out || !workspace || !a_rows || (N && !b_rows)) return NSA_EBADARG;
    const uint32_t N1 = N + (row_sums ? 1u : 0u);
    if (pairs < 1 || pairs > 2 || M < 1 || M > 64 || N1 < 1 || N > 192 || ld == 0 || ld % KSUB) return NSA_EBADARG;
    EmitGemmArgs g;
    g.emit = emit;
    g.ld = ld;
    for (uint32_t p = 0; p < 2; ++p) {
        g.a[p] = a_rows[p < pairs ? p : 0];
        g.b[p] = N ? b_rows[p < pairs ? p : 0] : 0;
    }
    g.pairs = pairs; g.M = M; g.N = N; g.sums = row_sums ? 1u : 0u;
    g.kw = emit_kw(ld);
    g.partial = workspace;
    const uint32_t slices = (uint32_t)((ld + g.kw - 1) / g.kw), tiles = (N + 31) / 32;
    launch_begin();
    if (M > 32) emit_launch<2>(g, slices, tiles, (hipStream_t)stream);
    else        emit_launch<1>(g, slices, tiles, (hipStream_t)stream);
    hipLaunchKernelGGL(k_emit_reduce, dim3((M * N1 + 31) / 32), dim3(256), 0, (hipStream_t)stream, workspace, out, M * N1, slices);
    return launch_end();
}

extern "C" uint64_t nsa_emit_gemm_workspace(uint64_t ld, uint32_t M, uint32_t N, int row_sums) {
    const uint32_t kw = nsa::emit_kw(ld);
    return kw ? ((ld + kw - 1) / kw) * (uint64_t)M * (N + (row_sums ? 1u : 0u)) : 0;
}
