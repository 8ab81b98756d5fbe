// Micro-test: is a VALU write to a v_mfma source operand (A or B) right after the MFMA's issue a WAR hazard on gfx950?
// One wave: acc = A*B with K nops between the MFMA and v_mov's that overwrite the operand registers with different data; the
// result must equal the un-disturbed product.  Fixed physical registers inside one asm block pin the instruction sequence.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

#define NOPS_0 ""
#define NOPS_1 "s_nop 0\n"
#define NOPS_2 "s_nop 1\n"
#define NOPS_4 "s_nop 3\n"
#define NOPS_8 "s_nop 7\n"
#define NOPS_16 "s_nop 15\n"

// SHAPE 32: v_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27]; SHAPE 16: v_mfma_f32_16x16x32_bf16 v[0:3], ...
// WHICH 0: overwrite A (v20..23), 1: overwrite B (v24..27), 2: nothing (reference)
#define KERNEL(NAME, MFMA, NACC, NOPS, OVERWRITE)                                                                      \
__global__ void NAME(const uint32_t* in, float* out) {                                                                 \
    const int l = threadIdx.x;                                                                                         \
    uint32_t a0 = in[l * 4 + 0], a1 = in[l * 4 + 1], a2 = in[l * 4 + 2], a3 = in[l * 4 + 3];                           \
    uint32_t b0 = in[256 + l * 4 + 0], b1 = in[256 + l * 4 + 1], b2 = in[256 + l * 4 + 2], b3 = in[256 + l * 4 + 3];   \
    uint32_t n0 = in[512 + l * 4 + 0], n1 = in[512 + l * 4 + 1], n2 = in[512 + l * 4 + 2], n3 = in[512 + l * 4 + 3];   \
    float o[16];                                                                                                       \
    asm volatile(                                                                                                      \
        "v_mov_b32 v20, %16\n v_mov_b32 v21, %17\n v_mov_b32 v22, %18\n v_mov_b32 v23, %19\n"                          \
        "v_mov_b32 v24, %20\n v_mov_b32 v25, %21\n v_mov_b32 v26, %22\n v_mov_b32 v27, %23\n"                          \
        "v_mov_b32 v30, %24\n v_mov_b32 v31, %25\n v_mov_b32 v32, %26\n v_mov_b32 v33, %27\n"                          \
        "v_mov_b32 v0, 0\n v_mov_b32 v1, 0\n v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n"   \
        "v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n" \
        "v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n"                                  \
        "s_nop 15\n s_nop 15\n"                                                                                        \
        MFMA "\n" NOPS OVERWRITE                                                                                       \
        "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"                                                                  \
        "v_mov_b32 %0, v0\n v_mov_b32 %1, v1\n v_mov_b32 %2, v2\n v_mov_b32 %3, v3\n v_mov_b32 %4, v4\n"               \
        "v_mov_b32 %5, v5\n v_mov_b32 %6, v6\n v_mov_b32 %7, v7\n v_mov_b32 %8, v8\n v_mov_b32 %9, v9\n"               \
        "v_mov_b32 %10, v10\n v_mov_b32 %11, v11\n v_mov_b32 %12, v12\n v_mov_b32 %13, v13\n v_mov_b32 %14, v14\n"     \
        "v_mov_b32 %15, v15\n"                                                                                         \
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),      \
          "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15]) \
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(n0), "v"(n1), "v"(n2), "v"(n3)   \
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v20", \
          "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v30", "v31", "v32", "v33", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", \
          "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65",   \
          "v66", "v67", "v68", "v69", "v70", "v71");                                \
    for (int r = 0; r < NACC; ++r) out[l * 16 + r] = o[r];                                                             \
}

#define M32 "v_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]"
#define M16 "v_mfma_f32_16x16x32_bf16 v[0:3], v[20:23], v[24:27], v[0:3]"
// the same with the matrix pipe busy: two independent MFMAs (accumulators v[40:55], v[56:71]; their own operand copies
// v[34:37], v[36:39]... are just the same A/B) issued right before the tested one
#define M32_BUSY "v_mfma_f32_32x32x16_bf16 v[40:55], v[20:23], v[24:27], v[40:55]\n v_mfma_f32_32x32x16_bf16 v[56:71], v[20:23], v[24:27], v[56:71]\n" M32
#define M16_BUSY "v_mfma_f32_16x16x32_bf16 v[40:43], v[20:23], v[24:27], v[40:43]\n v_mfma_f32_16x16x32_bf16 v[44:47], v[20:23], v[24:27], v[44:47]\n v_mfma_f32_16x16x32_bf16 v[48:51], v[20:23], v[24:27], v[48:51]\n" M16
#define OVA "v_mov_b32 v20, v30\n v_mov_b32 v21, v31\n v_mov_b32 v22, v32\n v_mov_b32 v23, v33\n"
#define OVB "v_mov_b32 v24, v30\n v_mov_b32 v25, v31\n v_mov_b32 v26, v32\n v_mov_b32 v27, v33\n"
#define OVB_REV "v_mov_b32 v27, v33\n v_mov_b32 v26, v32\n v_mov_b32 v25, v31\n v_mov_b32 v24, v30\n"

KERNEL(k32_ref, M32, 16, NOPS_0, "")
KERNEL(k32_a0, M32, 16, NOPS_0, OVA)  KERNEL(k32_a1, M32, 16, NOPS_1, OVA)  KERNEL(k32_a2, M32, 16, NOPS_2, OVA)
KERNEL(k32_a4, M32, 16, NOPS_4, OVA)  KERNEL(k32_a8, M32, 16, NOPS_8, OVA)
KERNEL(k32_b0, M32, 16, NOPS_0, OVB)  KERNEL(k32_b1, M32, 16, NOPS_1, OVB)  KERNEL(k32_b2, M32, 16, NOPS_2, OVB)
KERNEL(k32_b4, M32, 16, NOPS_4, OVB)  KERNEL(k32_b8, M32, 16, NOPS_8, OVB)  KERNEL(k32_br0, M32, 16, NOPS_0, OVB_REV)
KERNEL(k16_ref, M16, 4, NOPS_0, "")
KERNEL(k16_a0, M16, 4, NOPS_0, OVA)  KERNEL(k16_a1, M16, 4, NOPS_1, OVA)  KERNEL(k16_a2, M16, 4, NOPS_2, OVA)  KERNEL(k16_a4, M16, 4, NOPS_4, OVA)
KERNEL(k16_b0, M16, 4, NOPS_0, OVB)  KERNEL(k16_b1, M16, 4, NOPS_1, OVB)  KERNEL(k16_b2, M16, 4, NOPS_2, OVB)  KERNEL(k16_b4, M16, 4, NOPS_4, OVB)

KERNEL(k32q_ref, M32_BUSY, 16, NOPS_0, "")
KERNEL(k32q_a0, M32_BUSY, 16, NOPS_0, OVA)  KERNEL(k32q_a2, M32_BUSY, 16, NOPS_2, OVA)  KERNEL(k32q_a8, M32_BUSY, 16, NOPS_8, OVA)  KERNEL(k32q_a16, M32_BUSY, 16, NOPS_16, OVA)
KERNEL(k32q_b0, M32_BUSY, 16, NOPS_0, OVB)  KERNEL(k32q_b2, M32_BUSY, 16, NOPS_2, OVB)  KERNEL(k32q_b8, M32_BUSY, 16, NOPS_8, OVB)  KERNEL(k32q_b16, M32_BUSY, 16, NOPS_16, OVB)
KERNEL(k16q_ref, M16_BUSY, 4, NOPS_0, "")
KERNEL(k16q_a0, M16_BUSY, 4, NOPS_0, OVA)  KERNEL(k16q_a2, M16_BUSY, 4, NOPS_2, OVA)  KERNEL(k16q_a8, M16_BUSY, 4, NOPS_8, OVA)
KERNEL(k16q_b0, M16_BUSY, 4, NOPS_0, OVB)  KERNEL(k16q_b2, M16_BUSY, 4, NOPS_2, OVB)  KERNEL(k16q_b8, M16_BUSY, 4, NOPS_8, OVB)

typedef void (*kern_t)(const uint32_t*, float*);

int main() {
    std::vector<uint32_t> h(768);
    uint32_t s = 12345;
    auto rnd_bf16pair = [&]() {   // two bf16 values in [-1, 1)
        uint32_t w = 0;
        for (int k = 0; k < 2; ++k) {
            s = s * 1664525u + 1013904223u;
            float f = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f;
            uint32_t u; memcpy(&u, &f, 4);
            w |= (u >> 16) << (16 * k);
        }
        return w;
    };
    for (auto& v : h) v = rnd_bf16pair();
    uint32_t* din; float* dout;
    hipMalloc(&din, 768 * 4); hipMalloc(&dout, 64 * 16 * 4);
    hipMemcpy(din, h.data(), 768 * 4, hipMemcpyHostToDevice);
    struct T { const char* name; kern_t k; int nacc; int ref; };
    T tests[] = {{"32x32x16 ref", k32_ref, 16, -1},
                 {"32x32x16 overwrite A +0 nops", k32_a0, 16, 0}, {"32x32x16 overwrite A +1", k32_a1, 16, 0}, {"32x32x16 overwrite A +2", k32_a2, 16, 0},
                 {"32x32x16 overwrite A +4", k32_a4, 16, 0}, {"32x32x16 overwrite A +8", k32_a8, 16, 0},
                 {"32x32x16 overwrite B +0 nops", k32_b0, 16, 0}, {"32x32x16 overwrite B +1", k32_b1, 16, 0}, {"32x32x16 overwrite B +2", k32_b2, 16, 0},
                 {"32x32x16 overwrite B +4", k32_b4, 16, 0}, {"32x32x16 overwrite B +8", k32_b8, 16, 0}, {"32x32x16 overwrite B (v27 first) +0", k32_br0, 16, 0},
                 {"16x16x32 ref", k16_ref, 4, -1},
                 {"16x16x32 overwrite A +0 nops", k16_a0, 4, 12}, {"16x16x32 overwrite A +1", k16_a1, 4, 12}, {"16x16x32 overwrite A +2", k16_a2, 4, 12},
                 {"16x16x32 overwrite A +4", k16_a4, 4, 12},
                 {"16x16x32 overwrite B +0 nops", k16_b0, 4, 12}, {"16x16x32 overwrite B +1", k16_b1, 4, 12}, {"16x16x32 overwrite B +2", k16_b2, 4, 12},
                 {"16x16x32 overwrite B +4", k16_b4, 4, 12},
                 {"32x32x16 BUSY pipe ref", k32q_ref, 16, -1},
                 {"32x32x16 busy: overwrite A +0", k32q_a0, 16, 21}, {"32x32x16 busy: overwrite A +2", k32q_a2, 16, 21}, {"32x32x16 busy: overwrite A +8", k32q_a8, 16, 21},
                 {"32x32x16 busy: overwrite A +16", k32q_a16, 16, 21},
                 {"32x32x16 busy: overwrite B +0", k32q_b0, 16, 21}, {"32x32x16 busy: overwrite B +2", k32q_b2, 16, 21}, {"32x32x16 busy: overwrite B +8", k32q_b8, 16, 21},
                 {"32x32x16 busy: overwrite B +16", k32q_b16, 16, 21},
                 {"16x16x32 BUSY pipe ref", k16q_ref, 4, -1},
                 {"16x16x32 busy: overwrite A +0", k16q_a0, 4, 30}, {"16x16x32 busy: overwrite A +2", k16q_a2, 4, 30}, {"16x16x32 busy: overwrite A +8", k16q_a8, 4, 30},
                 {"16x16x32 busy: overwrite B +0", k16q_b0, 4, 30}, {"16x16x32 busy: overwrite B +2", k16q_b2, 4, 30}, {"16x16x32 busy: overwrite B +8", k16q_b8, 4, 30}};
    const int n = sizeof(tests) / sizeof(tests[0]);
    std::vector<std::vector<float>> res(n, std::vector<float>(64 * 16));
    for (int rep = 0; rep < 3; ++rep)
        for (int i = 0; i < n; ++i) {
            hipMemset(dout, 0, 64 * 16 * 4);
            hipLaunchKernelGGL(tests[i].k, dim3(1), dim3(64), 0, 0, din, dout);
            hipMemcpy(res[i].data(), dout, 64 * 16 * 4, hipMemcpyDeviceToHost);
            if (tests[i].ref < 0) { if (rep == 0) printf("%-40s checksum %.6f\n", tests[i].name, res[i][0] + res[i][17] + res[i][1000]); continue; }
            int bad = 0, first = -1, last = -1;
            unsigned long long lanes = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < tests[i].nacc; ++r)
                    if (res[i][l * 16 + r] != res[tests[i].ref][l * 16 + r]) { ++bad; lanes |= 1ull << l; if (first < 0) first = l; last = l; }
            if (rep == 0 || bad) printf("%-40s rep %d: %4d wrong accumulator values, lanes mask %016llx (first %d last %d)\n", tests[i].name, rep, bad, lanes, first, last);
        }
    return 0;
}
