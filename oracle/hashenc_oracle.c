/*
 * oracle/hashenc_oracle.c  --  TEST INFRASTRUCTURE ONLY (never on the product path).
 *
 * Plain-C, single-source CPU restatement of the reference's multi-resolution
 * hash/dense grid encoder: forward (+Jacobian), first backward (table scatter +
 * input gradient) and the "second backward" pair, as the reference's native
 * extension computes them:
 *     reference: code/hashencoder/src/hashencoder.cu
 *       fast_hash                               :35-51
 *       get_grid_index (dense / hashed switch)  :54-73
 *       smoothstep / smoothstep_derivative      :115-121
 *       kernel_grid                             :131-283
 *       kernel_grid_backward                    :286-373
 *       kernel_input_backward                   :376-402
 *       kernel_grid_second_backward_grad        :405-458
 *       kernel_grid_second_backward_embedding   :461-625
 *       host entry points (arg order, zeroing)  :758-854
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  It is the checker, not the thing shipped or measured.
 *
 * Parity pin: known-answer hash vectors + the reference's own
 * pure-torch twin HashEncoder.torch_forward (hashgrid.py:217-299) on dense
 * levels/interior points + goldens captured by running the reference's Python
 * wrappers (hashgrid.py:13-134) on top of this file (tests/golden/).
 *
 * Arithmetic notes (kept bit-for-bit deterministic; build with -ffp-contract=off):
 *   - all index arithmetic is uint32 with wrap-around exactly like the kernel:
 *     the dense stride is multiplied by `resolution` (not resolution+1) and a
 *     level whose stride wraps modulo 2^32 (resolution 2048, D=3: 2^33 -> 0)
 *     therefore takes the DENSE branch with a wrapped index (hashencoder.cu:60-70).
 *   - scale = exp2f(level*S)*H - 1 in float32, resolution = ceil(scale)+1 (:180-181).
 *   - per-(point,level) accumulation order follows the kernel (corner 0..2^D-1).
 *   - scatter order is sequential over points (the kernel's is atomic/unordered).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NSO_MAX_D 3u
#define NSO_MAX_C 8u

typedef struct {
    uint32_t row0;   /* first table row of the level (offsets[level])      */
    uint32_t rows;   /* hashmap_size = offsets[level+1]-offsets[level]     */
    uint32_t res;    /* resolution used for the dense stride                */
    float    scale;  /* x in [0,1] -> grid coordinate                       */
} nso_level_t;

static const uint32_t nso_primes[7] = {1u, 2654435761u, 805459861u, 3674653429u,
                                       2097192037u, 1434869437u, 2165219737u};

/* bound the OpenMP team (levels are the parallel axis: more threads than levels only adds spin-wait) */
void nso_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n < 1 ? 1 : n);
#else
    (void)n;
#endif
}

/* hashencoder.cu:35-51 */
uint32_t nso_fast_hash(const uint32_t *cell, uint32_t D) {
    uint32_t h = 0u;
    for (uint32_t d = 0; d < D; ++d) h ^= cell[d] * nso_primes[d];
    return h;
}

/* hashencoder.cu:54-73 -- returns the ROW inside the level (before *C + ch). */
uint32_t nso_level_row(uint32_t rows, uint32_t res, const uint32_t *cell, uint32_t D) {
    uint32_t stride = 1u, idx = 0u;
    for (uint32_t d = 0; d < D && stride <= rows; ++d) {
        idx += cell[d] * stride;
        stride *= res;
    }
    if (stride > rows) idx = nso_fast_hash(cell, D);
    return idx % rows;
}

/* hashencoder.cu:179-181 */
void nso_level_geometry(const int32_t *offsets, uint32_t level, float S, uint32_t H,
                        uint32_t *row0, uint32_t *rows, uint32_t *res, float *scale) {
    float sc = exp2f((float)level * S) * (float)H - 1.0f;
    *row0 = (uint32_t)offsets[level];
    *rows = (uint32_t)(offsets[level + 1] - offsets[level]);
    *scale = sc;
    *res = (uint32_t)ceilf(sc) + 1u;
}

static void level_geom(const int32_t *offsets, uint32_t level, float S, uint32_t H, nso_level_t *g) {
    nso_level_geometry(offsets, level, S, H, &g->row0, &g->rows, &g->res, &g->scale);
}

/* Range test + cell/fraction split shared by all kernels
 * (hashencoder.cu:152-159, 188-195). Returns 0 when the point is out of [0,1]^D. */
static int locate(const float *x, uint32_t D, float scale, uint32_t *cell, float *w, float *dw) {
    for (uint32_t d = 0; d < D; ++d)
        if (x[d] < 0.0f || x[d] > 1.0f) return 0;
    for (uint32_t d = 0; d < D; ++d) {
        float p = x[d] * scale;
        cell[d] = (uint32_t)floorf(p);
        float t = p - (float)cell[d];
        dw[d] = 6.0f * t * (1.0f - t);            /* smoothstep'  :119-121 */
        w[d] = t * t * (3.0f - 2.0f * t);         /* smoothstep   :115-117 */
    }
    return 1;
}

/* ---------------------------------------------------------------- forward */
/* inputs[B,D] in [0,1]; emb[rows_total,C]; offsets[L+1]; outputs[L,B,C];
 * dy_dx[B,L,D,C] (written only when calc_grad_inputs).  hashencoder.cu:131-283 */
int nso_hash_encode_forward(const float *inputs, const float *emb, const int32_t *offsets,
                            float *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                            float S, uint32_t H, int calc_grad_inputs, float *dy_dx) {
    if (D < 2 || D > NSO_MAX_D) return 1;
    if (!(C == 1 || C == 2 || C == 4 || C == 8)) return 1;
#pragma omp parallel for schedule(static)
    for (int64_t lv = 0; lv < (int64_t)L; ++lv) {
        nso_level_t g;
        level_geom(offsets, (uint32_t)lv, S, H, &g);
        const float *tab = emb + (size_t)g.row0 * C;
        for (uint32_t b = 0; b < B; ++b) {
            const float *x = inputs + (size_t)b * D;
            float *out = outputs + ((size_t)lv * B + b) * C;
            float *jac = calc_grad_inputs ? dy_dx + (((size_t)b * L + lv) * D) * C : 0;
            uint32_t cell[NSO_MAX_D];
            float w[NSO_MAX_D], dw[NSO_MAX_D];
            if (!locate(x, D, g.scale, cell, w, dw)) {
                for (uint32_t c = 0; c < C; ++c) out[c] = 0.0f;
                if (jac) for (uint32_t i = 0; i < D * C; ++i) jac[i] = 0.0f;
                continue;
            }
            float acc[NSO_MAX_C] = {0};
            for (uint32_t corner = 0; corner < (1u << D); ++corner) {
                float wt = 1.0f;
                uint32_t q[NSO_MAX_D];
                for (uint32_t d = 0; d < D; ++d) {
                    if (corner & (1u << d)) { wt *= w[d]; q[d] = cell[d] + 1u; }
                    else                    { wt *= 1.0f - w[d]; q[d] = cell[d]; }
                }
                const float *v = tab + (size_t)nso_level_row(g.rows, g.res, q, D) * C;
                for (uint32_t c = 0; c < C; ++c) acc[c] += wt * v[c];
            }
            for (uint32_t c = 0; c < C; ++c) out[c] = acc[c];
            if (!jac) continue;
            /* Jacobian d out / d x[gd]  :239-282 */
            for (uint32_t gd = 0; gd < D; ++gd) {
                float jacc[NSO_MAX_C] = {0};
                for (uint32_t face = 0; face < (1u << (D - 1)); ++face) {
                    float wt = g.scale;
                    uint32_t q[NSO_MAX_D];
                    for (uint32_t nd = 0; nd < D - 1; ++nd) {
                        uint32_t d = (nd >= gd) ? nd + 1 : nd;
                        if (face & (1u << nd)) { wt *= w[d]; q[d] = cell[d] + 1u; }
                        else                   { wt *= 1.0f - w[d]; q[d] = cell[d]; }
                    }
                    q[gd] = cell[gd];
                    const float *lo = tab + (size_t)nso_level_row(g.rows, g.res, q, D) * C;
                    q[gd] = cell[gd] + 1u;
                    const float *hi = tab + (size_t)nso_level_row(g.rows, g.res, q, D) * C;
                    for (uint32_t c = 0; c < C; ++c) jacc[c] += wt * (hi[c] - lo[c]) * dw[gd];
                }
                for (uint32_t c = 0; c < C; ++c) jac[gd * C + c] = jacc[c];
            }
        }
    }
    return 0;
}

/* --------------------------------------------------------- first backward */
/* grad[L,B,C]; grad_emb[rows_total,C] is accumulated INTO (caller pre-zeroes);
 * grad_inputs[B,D] is overwritten when calc_grad_inputs.
 * hashencoder.cu:286-402, launch logic :655-680 */
int nso_hash_encode_backward(const float *grad, const float *inputs, const float *emb,
                             const int32_t *offsets, float *grad_emb, uint32_t B, uint32_t D,
                             uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                             const float *dy_dx, float *grad_inputs) {
    (void)emb;
    if (D < 2 || D > NSO_MAX_D) return 1;
    if (!(C == 1 || C == 2 || C == 4 || C == 8)) return 1;
#pragma omp parallel for schedule(static)
    for (int64_t lv = 0; lv < (int64_t)L; ++lv) {
        nso_level_t g;
        level_geom(offsets, (uint32_t)lv, S, H, &g);
        float *gtab = grad_emb + (size_t)g.row0 * C;
        for (uint32_t b = 0; b < B; ++b) {
            uint32_t cell[NSO_MAX_D];
            float w[NSO_MAX_D], dw[NSO_MAX_D];
            if (!locate(inputs + (size_t)b * D, D, g.scale, cell, w, dw)) continue;
            const float *gy = grad + ((size_t)lv * B + b) * C;
            for (uint32_t corner = 0; corner < (1u << D); ++corner) {
                float wt = 1.0f;
                uint32_t q[NSO_MAX_D];
                for (uint32_t d = 0; d < D; ++d) {
                    if (corner & (1u << d)) { wt *= w[d]; q[d] = cell[d] + 1u; }
                    else                    { wt *= 1.0f - w[d]; q[d] = cell[d]; }
                }
                float *dst = gtab + (size_t)nso_level_row(g.rows, g.res, q, D) * C;
                for (uint32_t c = 0; c < C; ++c) dst[c] += wt * gy[c];
            }
        }
    }
    if (calc_grad_inputs) {
        /* grad_inputs[b,d] = sum_l sum_c grad[l,b,c]*dy_dx[b,l,d,c]   :376-402 */
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < (int64_t)B; ++b) {
            for (uint32_t d = 0; d < D; ++d) {
                float r = 0.0f;
                for (uint32_t l = 0; l < L; ++l)
                    for (uint32_t c = 0; c < C; ++c)
                        r += grad[((size_t)l * B + b) * C + c] *
                             dy_dx[(((size_t)b * L + l) * D + d) * C + c];
                grad_inputs[(size_t)b * D + d] = r;
            }
        }
    }
    return 0;
}

/* -------------------------------------------------------- second backward */
/* grad_grad[L,B,C] overwritten with J.ggi; grad2_emb accumulated INTO (pre-zeroed).
 * The derivative w.r.t. the inputs is NOT produced (hashgrid.py:134 returns None).
 * hashencoder.cu:405-625, launch logic :696-735 (C==1 is rejected there). */
int nso_hash_encode_second_backward(const float *grad, const float *inputs, const float *emb,
                                    const int32_t *offsets, uint32_t B, uint32_t D, uint32_t C,
                                    uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                                    const float *dy_dx, const float *grad_grad_inputs,
                                    float *grad_grad, float *grad2_emb) {
    (void)emb; (void)calc_grad_inputs;
    if (D < 2 || D > NSO_MAX_D) return 1;
    if (!(C == 2 || C == 4 || C == 8)) return 1;
#pragma omp parallel for schedule(static)
    for (int64_t lv = 0; lv < (int64_t)L; ++lv) {
        nso_level_t g;
        level_geom(offsets, (uint32_t)lv, S, H, &g);
        float *gtab = grad2_emb + (size_t)g.row0 * C;
        for (uint32_t b = 0; b < B; ++b) {
            const float *ggi = grad_grad_inputs + (size_t)b * D;
            const float *jac = dy_dx + (((size_t)b * L + lv) * D) * C;
            float *gg = grad_grad + ((size_t)lv * B + b) * C;
            /* :405-440  (no range test: dy_dx is zero for out-of-range points) */
            for (uint32_t c = 0; c < C; ++c) {
                float r = 0.0f;
                for (uint32_t d = 0; d < D; ++d) r += ggi[d] * jac[d * C + c];
                gg[c] = r;
            }
            /* :461-625 */
            uint32_t cell[NSO_MAX_D];
            float w[NSO_MAX_D], dw[NSO_MAX_D];
            if (!locate(inputs + (size_t)b * D, D, g.scale, cell, w, dw)) continue;
            const float *gy = grad + ((size_t)lv * B + b) * C;
            float cache[(1u << NSO_MAX_D) * NSO_MAX_C];
            memset(cache, 0, sizeof(cache));
            for (uint32_t gd = 0; gd < D; ++gd) {
                for (uint32_t face = 0; face < (1u << (D - 1)); ++face) {
                    float wt = g.scale;
                    uint32_t lo = 0u;
                    for (uint32_t nd = 0; nd < D - 1; ++nd) {
                        uint32_t d = (nd >= gd) ? nd + 1 : nd;
                        if (face & (1u << nd)) { wt *= w[d]; lo |= 1u << d; }
                        else                   { wt *= 1.0f - w[d]; }
                    }
                    uint32_t hi = lo | (1u << gd);
                    for (uint32_t c = 0; c < C; ++c) {
                        float v = wt * gy[c] * ggi[gd] * dw[gd];
                        cache[hi * C + c] += v;
                        cache[lo * C + c] -= v;
                    }
                }
            }
            for (uint32_t corner = 0; corner < (1u << D); ++corner) {
                uint32_t q[NSO_MAX_D];
                for (uint32_t d = 0; d < D; ++d) q[d] = cell[d] + ((corner >> d) & 1u);
                float *dst = gtab + (size_t)nso_level_row(g.rows, g.res, q, D) * C;
                for (uint32_t c = 0; c < C; ++c) dst[c] += cache[corner * C + c];
            }
        }
    }
    return 0;
}
