/*
 * nicer_slam_amd.h -- C ABI of the MI355X-native NICER-SLAM render core (libnicer_slam_amd.so).
 *
 * Plain pointers and sizes only; every buffer is DEVICE memory owned by the caller unless the
 * parameter name ends in `_host`.  Nothing is allocated or kept between calls.  All launches are
 * asynchronous on `stream` (a hipStream_t; NULL = the legacy default stream, which is what the
 * reference launches on).  Every entry point returns 0 (NSA_OK) or an NSA_E* code; the text the
 * reference would have thrown for that condition is available from nsa_strerror().
 *
 * Section 1 replaces, one for one, the three functions of the reference's native module
 * (reference: code/hashencoder/src/hashencoder.h:13-15, bound at code/hashencoder/src/bindings.cpp:5-7,
 *  implemented at code/hashencoder/src/hashencoder.cu:758-854).  Differences from that interface:
 *   - raw device pointers + an explicit stream instead of at::Tensor (the tensor checks of
 *     hashencoder.cu:759-775 live in the thin binding, nicer_slam_amd/hashencoder/backend.py);
 *   - `offsets_host` is a HOST copy of the int32 offsets tensor (the per-level geometry is derived
 *     on the host and passed as kernel arguments, so the device never calls exp2f/ceil);
 *   - float32 only (the reference also dispatches half/double; it never runs them);
 *   - an int status instead of a C++ exception.
 * Layouts are the reference's: inputs[B,D] in [0,1]; embeddings[rows,C]; outputs[L,B,C] (level-major);
 * dy_dx[B,L,D,C]; grad[L,B,C]; grad_embeddings/grad2_embeddings[rows,C] are ACCUMULATED INTO with
 * float atomics (caller pre-zeroes, as hashgrid.py:84-85,117-118 does); grad_inputs[B,D] and
 * grad_grad[L,B,C] are overwritten.
 */
#ifndef NICER_SLAM_AMD_H
#define NICER_SLAM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSA_OK 0
#define NSA_EUNSUPPORTED_C 1   /* "GridEncoding: C must be 1, 2, 4, or 8." (hashencoder.cu:637,652,...) */
#define NSA_ETOO_MANY_LEVELS 2 /* more than NSA_MAX_LEVELS levels */
#define NSA_ELAUNCH 3          /* hipGetLastError() != hipSuccess after a launch */
#define NSA_EBADARG 4          /* null pointer / inconsistent sizes */
#define NSA_EUNSUPPORTED_NET 5 /* fused core: network shape outside the compiled set */

#define NSA_MAX_LEVELS 32

typedef void *nsa_stream_t; /* hipStream_t */

const char *nsa_strerror(int code);
int nsa_version(void);

/* ---- Section 1: hash/dense multi-resolution grid encoder (reference hashencoder.h:13-15) ---- */

/* replaces hash_encode_forward (hashencoder.cu:758-781 -> kernel_grid :131-283).
 * dy_dx may be NULL when calc_grad_inputs == 0. */
int nsa_hash_encode_forward(const float *inputs, const float *embeddings, const int32_t *offsets_host,
                            float *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                            uint32_t H, int calc_grad_inputs, float *dy_dx, nsa_stream_t stream);

/* replaces hash_encode_backward (hashencoder.cu:783-813 -> kernel_grid_backward :286-373 and
 * kernel_input_backward :376-402).  grad_embeddings may be NULL to skip the table scatter
 * (extension: used when the table does not require grad). */
int nsa_hash_encode_backward(const float *grad, const float *inputs, const float *embeddings,
                             const int32_t *offsets_host, float *grad_embeddings, uint32_t B, uint32_t D,
                             uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                             const float *dy_dx, float *grad_inputs, nsa_stream_t stream);

/* replaces hash_encode_second_backward (hashencoder.cu:816-854 -> kernel_grid_second_backward_grad
 * :405-458 and kernel_grid_second_backward_embedding :461-625).  C == 1 is rejected like the
 * reference (:708-714).  grad2_embeddings may be NULL to skip the table scatter (extension). */
int nsa_hash_encode_second_backward(const float *grad, const float *inputs, const float *embeddings,
                                    const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C,
                                    uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                                    const float *dy_dx, const float *grad_grad_inputs, float *grad_grad,
                                    float *grad2_embeddings, nsa_stream_t stream);

/* ---- Section 2: fused render core (no native counterpart in the reference: these replace the PyTorch-level
 * functions named at each entry point; rays, samples and per-point quantities stay in HBM between them) ---- */

/* One multi-resolution grid + the MLP that consumes it. */
typedef struct nsa_grid {
    const float *table;          /* embeddings[rows, C] (device)                                   */
    const int32_t *offsets_host; /* [L+1] (host)                                                   */
    uint32_t L, C;               /* levels, features per level                                     */
    float S;                     /* log2(per_level_scale)                                          */
    uint32_t H;                  /* base resolution                                                */
    float divide_factor;         /* x is divided by this before the [-1,1] -> [0,1] map            */
    uint32_t n_hidden;           /* hidden layers of the attached MLP (coarse 1, fine 3)           */
    uint32_t precision;          /* GEMM operands of the attached MLP: 0 = fp32-faithful (3-way bf16 split, the
                                  * reference's precision; default), 1 = plain bf16 operands with fp32 accumulation
                                  * (optional "bf16 MLP" mode of BASELINE configs 2/4; encoders, activations,
                                  * compositing and all reductions stay fp32)                       */
    uint32_t tile;               /* tiling of the SDF-network kernels and the matching packed-parameter layout: 16 = quad
                                  * tiling (16 points per wave, four lanes per point; pack_sdf_net4), 0 or 32 = 32-point
                                  * tiling (lane pair per point; pack_sdf_net).  Ignored by the colour network.          */
} nsa_grid_t;

/* Where the points of a per-point kernel come from: sample (pid % S) of ray (pid / S), x = o + z d -- or, when
 * `points` is non-NULL, an explicit list (eikonal samples). */
typedef struct nsa_points {
    const float *rays_o;  /* [R,3] */
    const float *rays_d;  /* [R,3] */
    const float *z_vals;  /* [R,S] */
    const float *points;  /* [P,3] or NULL */
    uint32_t P, S;        /* P = R*S in ray mode */
    const uint32_t *order; /* optional [P] launch order (a permutation): work item i processes point order[i].  Per-point
                            * arrays (sdf, grad, rgb, g_*) stay indexed by point; tile-indexed buffers (HL feature
                            * vectors, the colour save area, emission rows) are indexed by work item, consistently
                            * across the forward and backward entry points.  NULL = identity. */
} nsa_points_t;

/* Per-point feature vectors travel in "HL" layout (the MFMA register image): float index ((tile*32+q)*64+lane),
 * tile = point/32, sized ceil(P/32)*2048 floats; see csrc/mlp_common.hpp. */

/* One SDF network (coarse or fine) at P points: sdf, grad sdf (reverse pass, kept differentiable by
 * nsa_sdfnet_backward) and the 64-feature vector.  accumulate != 0 adds to sdf/grad/feat (the fine network on
 * top of the coarse one).  replaces ImplicitNetworkGrid.get_outputs / ImplicitNetworkGrid_COMBINE.get_outputs
 * (code/model/base_networks.py:34-40,208-221) incl. HashEncoder.forward and the positional encoding. */
int nsa_sdfnet_forward(const nsa_points_t *pts, const nsa_grid_t *grid, const float *packed, int accumulate,
                       float *sdf, float *grad, float *feat_hl, nsa_stream_t stream);

/* Both networks of the COMBINE in one launch: what nsa_sdfnet_forward(coarse, accumulate 0) followed by
 * nsa_sdfnet_forward(fine, accumulate 1) leaves in sdf / grad / feat_hl, bit for bit, for the quad tiling (both descriptors
 * tile == 16 with their quad packs, the same precision).  Point, positional encoding, level geometry and outputs are handled
 * once; the coarse results stay in registers.  replaces ImplicitNetworkGrid_COMBINE.get_outputs
 * (code/model/base_networks.py:7-47) in the "fine" stage. */
int nsa_sdfnet_forward_pair(const nsa_points_t *pts, const nsa_grid_t *coarse, const nsa_grid_t *fine,
                            const float *packed_coarse, const float *packed_fine, float *sdf, float *grad, float *feat_hl,
                            nsa_stream_t stream);

/* Backward of the above for the DATA path: given d/d(sdf)[P], d/d(feat) (HL), d/d(grad sdf)[P,3] (any may be
 * NULL = zero) produce d/dx [P,3] -- value path + double backward through the reverse pass, with exactly the terms
 * of the reference graph (the grid-Hessian term is dropped, code/hashencoder/hashgrid.py:134). */
int nsa_sdfnet_backward(const nsa_points_t *pts, const nsa_grid_t *grid, const float *packed, const float *g_sdf,
                        const float *g_feat_hl, const float *g_grad, int accumulate, float *g_x,
                        nsa_stream_t stream);

/* Pixels -> rays for b images x n pixels: rays_o[b*n,3] = pose[:3,3], rays_d = (p - o)/|p - o|^2 (NOT unit length),
 * depth_scale[b*n] = z of the identity-pose direction.  replaces rend_util.get_camera_params + lift
 * (code/utils/rend_util.py:68-93,107-129), both calls of code/model/network.py:98-102. */
int nsa_rays_forward(const float *uv, const float *pose, const float *K, uint32_t b, uint32_t n, float *rays_o,
                     float *rays_d, float *depth_scale, nsa_stream_t stream);

/* nsa_rays_forward and the sampler's draws of the pass (nsa_draw: n_rand uniforms into t_rand, the n_extra picks of E into
 * extra_idx, no eikonal picks) in ONE launch -- the draw's workgroups ride beside the ray lifting.  Same rays, same draws. */
int nsa_rays_forward_draw(const float *uv, const float *pose, const float *K, uint32_t b, uint32_t n, float *rays_o,
                          float *rays_d, float *depth_scale, uint64_t *state, uint64_t n_rand, float *t_rand, uint32_t E,
                          uint32_t n_extra, uint32_t S, int32_t *extra_idx, nsa_stream_t stream);

/* Backward of the above to the camera-to-world matrices: g_pose[b,4,4] (overwritten; bottom row zero).  Deterministic: one
 * workgroup per image, fixed-order sums, no atomics. */
int nsa_rays_pose_backward(const float *uv, const float *pose, const float *K, uint32_t b, uint32_t n,
                           const float *g_rays_o, const float *g_rays_d, float *g_pose, nsa_stream_t stream);

/* Colour network at the composite points: rgb = sigmoid(MLP([x, PE4(view dir), grad sdf, feature, colour grid])).
 * replaces RenderingNetwork.forward, mode "idr" (code/model/base_networks.py:333-395).  `save` (optional,
 * ceil(P/32)*(4096 + 256) floats) receives what the backward needs: from the 1 GiB colour table the features + Jacobian (4096 floats per
 * 32-point tile), and of the MLP itself the ReLU masks of both hidden layers and the sigmoid outputs (256 floats per tile, behind the
 * ceil(P/32)*4096 block) -- the data-path backward (nsa_colour_backward, nsa_colour_coarse_backward) reads those instead of
 * recomputing the forward; the mapping backward (nsa_colour_backward_params) recomputes it (its activations are gradient operands). */
int nsa_colour_forward(const nsa_points_t *pts, const nsa_grid_t *grid, const float *packed, const float *grad,
                       const float *feat_hl, float *rgb, float *save, nsa_stream_t stream);

/* nsa_colour_forward followed by nsa_composite_forward as two phases of ONE launch, under the conditions of
 * nsa_colour_forward_track below (ray samples in ray order, 128 per ray: a workgroup of the colour forward is one ray): weights,
 * rgb_values, depth, nmap, entropy as nsa_composite_forward leaves them.  Identical results. */
int nsa_colour_forward_composite(const nsa_points_t *pts, const nsa_grid_t *grid, const float *packed, const float *grad,
                                 const float *feat_hl, float *rgb, float *save, const float *sdf, const float *voxels,
                                 uint32_t voxel_res, float *weights, float *rgb_values, float *depth, float *nmap,
                                 float *entropy, nsa_stream_t stream);

/* nsa_colour_forward followed by nsa_composite_track as two phases of ONE launch, for ray samples in ray order with 128 samples per
 * ray (P a multiple of 128, no launch order): a workgroup of the colour forward holds exactly one ray, and when its colours are stored
 * its first wave forms the ray's rendered colour, the L1 cotangent and the composite backward (arguments as for nsa_composite_track;
 * R = P / 128).  The same statements as the two kernels: identical results; the per-ray kernel's launch disappears into the tail of
 * the colour forward.  This entry leaves the 16 FEATURE slots per lane of `save` unwritten (the Jacobian, the ReLU masks and the outputs
 * are written): what follows it is the data-path backward; nsa_colour_backward_params needs a save area written by nsa_colour_forward
 * or nsa_colour_forward_composite. */
int nsa_colour_forward_track(const nsa_points_t *pts, const nsa_grid_t *grid, const float *packed, const float *grad,
                             const float *feat_hl, float *rgb, float *save, const float *sdf, const float *voxels,
                             uint32_t voxel_res, const float *gt, uint32_t n_total, float *rgb_values, float *ray_loss,
                             float *g_sdf, float *g_rgb, float *g_grad, nsa_stream_t stream);

/* Data-path backward of the colour network: d/d(rgb)[P,3] -> d/d(feat) (HL, overwritten), d/d(grad sdf)[P,3]
 * (ADDED into g_grad), d/dx [P,3] and d/d(view dir) [P,3] (overwritten).  grid_grad = 0 reproduces
 * color_stage == "base" (grid feature detached, base_networks.py:337-339). */
int nsa_colour_backward(const nsa_points_t *pts, const nsa_grid_t *grid, const float *packed, const float *grad,
                        const float *feat_hl, const float *save, const float *g_rgb, int grid_grad,
                        float *g_feat_hl, float *g_grad, float *g_x, float *g_dir, nsa_stream_t stream);

/* nsa_colour_backward followed by nsa_sdfnet_backward(coarse, accumulate 1) -- the coarse network in the 32-point tiling, the
 * tiles of the colour kernels -- as two phases of ONE launch: every wave runs the colour backward of its 32 points and then the
 * coarse SDF backward of the same points (g_sdf: d/d sdf from the composite; the feature and normal cotangents and the d/dx to
 * accumulate onto are what its first phase has just written).  The same statements as the two kernels: identical results.
 * For a tracking batch, where both kernels run only two rounds of waves, the input burst of a round is paid once. */
int nsa_colour_coarse_backward(const nsa_points_t *pts, const nsa_grid_t *grid, const float *packed, const float *grad,
                               const float *feat_hl, const float *save, const float *g_rgb, int grid_grad, float *g_feat_hl,
                               float *g_grad, float *g_x, float *g_dir, const nsa_grid_t *coarse, const float *packed_coarse,
                               const float *g_sdf, nsa_stream_t stream);

/* Mapping-mode backward (parameter gradients): the same kernels as nsa_sdfnet_backward / nsa_colour_backward with two
 * more outputs.  replaces, for a mapping iteration, what torch.autograd does through ImplicitNetworkGrid /
 * RenderingNetwork / _hash_encode.backward / _hash_encode_second_backward (code/model/base_networks.py:195-221,
 * 333-395; code/hashencoder/hashgrid.py:64-141) for the trainable parameters of volsdf_train.py:150-173:
 *   g_table  gradient of the grid table (same shape as grid->table), ATOMICALLY ACCUMULATED (caller zeroes it):
 *            value path + (SDF grids) the table's share of the double backward through grad sdf; may be NULL.
 *   emit     [nsa_*_emit_rows()][emit_ld] per-point vectors, column = point index (emit_ld >= ceil(P/32)*32 -- the quad tiling writes 16-point tiles, so its padding starts at ceil(P/16)*16 --, columns
 *            of padding points are written as 0).  The weight gradients are GEMMs over these rows (row map: the
 *            SE_* / CE_* enums in csrc/render_sdfnet.hip, csrc/render_colour.hip; host side fused/mapping.py).
 *            SDF: nsa_sdfnet_emit_rows_nh(grid->n_hidden) rows (464 for the coarse network, 976 for the fine one, whose
 *            gradients the reference computes but never applies, volsdf_train.py:150-173); may be NULL. */
int nsa_sdfnet_backward_params(const nsa_points_t *pts, const nsa_grid_t *grid, const float *packed, const float *g_sdf,
                               const float *g_feat_hl, const float *g_grad, int accumulate, float *g_x, float *g_table,
                               float *emit, uint32_t emit_ld, nsa_stream_t stream);
int nsa_colour_backward_params(const nsa_points_t *pts, const nsa_grid_t *grid, const float *packed, const float *grad,
                               const float *feat_hl, const float *save, const float *g_rgb, int grid_grad,
                               float *g_feat_hl, float *g_grad, float *g_x, float *g_dir, float *g_table, float *emit,
                               uint32_t emit_ld, nsa_stream_t stream);
int nsa_sdfnet_emit_rows(void);                 /* one hidden layer (coarse network) */
int nsa_sdfnet_emit_rows_nh(uint32_t n_hidden); /* 1 or 3 hidden layers; -1 otherwise (32-point tiling) */
int nsa_sdfnet_emit_rows_tile(uint32_t n_hidden, uint32_t tile); /* the same for nsa_grid_t.tile = 16 / 32 */
int nsa_colour_emit_rows(void);

/* Per-ray SDF -> density -> alpha compositing.  replaces SLAMNetwork.volume_rendering (code/model/network.py:349-370)
 * + the composite sums of SLAMNetwork.forward (:147-151, 298, 338-342) + GridPredefineDensity (density.py:37-67).
 * Outputs weights[R,S], rgb_values[R,3], depth[R] (= sum w z / (sum w + 1e-8), before depth_scale),
 * nmap[R,3] (= sum w grad/(|grad|+1e-6), before the rotation by the pose), entropy[R] (= sum -w log(w+1e-4)). */
int nsa_composite_forward(const float *rays_o, const float *rays_d, const float *z_vals, const float *sdf,
                          const float *rgb, const float *grad, const float *voxels, uint32_t voxel_res, uint32_t R,
                          uint32_t S, float *weights, float *rgb_values, float *depth, float *nmap, float *entropy,
                          nsa_stream_t stream);

/* Backward of the above: per-ray cotangents (any may be NULL) -> d/d(sdf)[R,S], d/d(rgb)[R,S,3], d/d(grad)[R,S,3]. */
int nsa_composite_backward(const float *rays_o, const float *rays_d, const float *z_vals, const float *sdf,
                           const float *rgb, const float *grad, const float *voxels, uint32_t voxel_res, uint32_t R,
                           uint32_t S, const float *g_rgb_values, const float *g_depth, const float *g_nmap,
                           const float *g_entropy, const float *g_weights, float *g_sdf, float *g_rgb, float *g_grad,
                           nsa_stream_t stream);

/* x = o + z d, view dir = d:  d/d(rays_o)[R,3] = sum_i g_x ;  d/d(rays_d)[R,3] = sum_i z_i g_x + sum_i g_dir. */
int nsa_rays_backward(const float *z_vals, const float *g_x, const float *g_dir, uint32_t R, uint32_t S,
                      float *g_rays_o, float *g_rays_d, nsa_stream_t stream);

/* Coarse sampler stage: stratified z on [near, cube exit], points, coarse+fine SDF at R*E points (no grad).
 * replaces UniformSampler.get_z_vals (code/model/ray_sampler.py:37-61) + ImplicitNetworkGrid_COMBINE.get_sdf_vals
 * (code/model/base_networks.py:27-32) as called from ImportantSampler.get_z_vals (ray_sampler.py:92-102).
 * t_lin = linspace(0,1,E); t_rand = per-sample jitter in [0,1) or NULL (eval mode); packed_* = MLP parameters
 * in MFMA fragment order (nicer_slam_amd/fused/pack.py).  Outputs z[R,E], sdf[R,E], far[R]. */
int nsa_sampler_sdf(const float *rays_o, const float *rays_d, uint32_t R, uint32_t E, const float *t_lin,
                    const float *t_rand, float near, float bound, float far_cap, const nsa_grid_t *coarse,
                    const nsa_grid_t *fine, const float *packed_coarse, const float *packed_fine, float *z,
                    float *sdf, float *far, nsa_stream_t stream);

/* Per-ray importance stage: density -> weights -> cdf -> N inverse-CDF samples, merged with near, far and
 * n_extra of the coarse samples, sorted.  replaces ImportantSampler.get_z_vals (ray_sampler.py:104-159) and
 * GridPredefineDensity (code/model/density.py:37-67).  z_vals[R, N+2+n_extra]; z_eik[R] = z_vals[r, eik_idx[r]]
 * (optional).  One workgroup per ray; N+2+n_extra <= 256, voxel_res <= 1024. */
int nsa_sample_rays(const float *rays_o, const float *rays_d, const float *z, const float *sdf, const float *far,
                    const float *voxels, uint32_t voxel_res, uint32_t R, uint32_t E, uint32_t N, const float *u_lin,
                    const int32_t *extra_idx, uint32_t n_extra, float near, const int32_t *eik_idx, float *z_vals,
                    float *z_eik, nsa_stream_t stream);

/* ---- Section 3: scalar head/tail of a tracking iteration (so a whole iteration is a fixed kernel sequence) ---- */

/* cam[b,7] = (qw,qx,qy,qz,tx,ty,tz) -> pose[b,4,4].  replaces quad2rotation / get_camera_from_tensor
 * (code/utils/general.py:52-100); nsa_pose_grad_to_cam is its backward (g_pose[b,4,4] -> g_cam[b,7]). */
int nsa_cam_to_pose(const float *cam, uint32_t b, float *pose, nsa_stream_t stream);
int nsa_pose_grad_to_cam(const float *cam, const float *g_pose, uint32_t b, float *g_cam, nsa_stream_t stream);

/* loss[0] = mean |pred - target| over n scalars, g_pred = sign(pred - target)/n.  replaces SLAMLoss.get_rgb_loss with
 * torch.nn.L1Loss(reduction="mean") (code/model/loss.py:57-65,131) and its backward. */
int nsa_l1_loss(const float *pred, const float *target, uint32_t n, float *loss, float *g_pred, nsa_stream_t stream);

/* torch.optim.Adam step (no weight decay / amsgrad) on n <= 256 parameters; `step` is a device scalar that is
 * incremented; lr_step > 0 applies StepLR(lr_step, lr_gamma).  replaces optimizer_camera.step() +
 * scheduler_camera.step() (code/training/volsdf_train.py:396-399,425-427). */
int nsa_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *step, uint32_t n,
                  float lr, float beta1, float beta2, float eps, uint32_t lr_step, float lr_gamma,
                  nsa_stream_t stream);

/* Integer draws of one sampler call from a buffer u of E + R uniforms in [0,1): extra_idx[n_extra] = the first n_extra
 * entries of a random permutation of 0..E-1 (E <= 1024), eik_idx[R] = uniform in 0..S-1; either output may be NULL.
 * replaces torch.randperm(E)[:n_extra] and torch.randint(S, (R,)) (code/model/ray_sampler.py:148,158). */
int nsa_draw_picks(const float *u, uint32_t E, uint32_t n_extra, uint32_t R, uint32_t S, int32_t *extra_idx,
                   int32_t *eik_idx, nsa_stream_t stream);

/* Every random draw of one sampler call in one launch, from a counter-based generator (Philox4x32-10, the generator behind
 * torch.rand) whose state the caller owns: state = 4 x uint64 on the device {seed, call number, 0, 0}; the kernel advances the
 * call number itself, so a captured graph draws fresh numbers on every replay.  t_rand[n_rand] uniforms in [0,1) (the stratified
 * jitter, ray_sampler.py:57-58), extra_idx / eik_idx as nsa_draw_picks (any output may be NULL).  Value i of call c is
 * philox(key = seed, counter = (i / 4, region, c))[i % 4] >> 8, region 0 = t_rand, 1 = the E permutation keys, 2 = eik_idx.
 * replaces torch.rand / torch.randperm / torch.randint of code/model/ray_sampler.py:57-58,148,158. */
int nsa_draw(uint64_t *state, uint64_t n_rand, float *t_rand, uint32_t E, uint32_t n_extra, uint32_t R, uint32_t S,
             int32_t *extra_idx, int32_t *eik_idx, nsa_stream_t stream);

/* SDF (coarse + fine; fine == NULL: stage "coarse") at N explicit points, no gradients: batch inference for mesh
 * extraction grids and plots.  replaces ImplicitNetworkGrid_COMBINE.get_sdf_vals (code/model/base_networks.py:25-35)
 * as called by code/utils/plots.py:91,142. */
int nsa_sdf_points(const float *points, uint64_t N, const nsa_grid_t *coarse, const nsa_grid_t *fine,
                   const float *packed_coarse, const float *packed_fine, float *sdf, nsa_stream_t stream);

/* Fused head and tail of a single-image tracking iteration (graph-captured tracker): fewer, larger graph nodes.
 *   nsa_track_head = nsa_cam_to_pose + nsa_rays_forward (b = 1);
 *   nsa_track_tail = nsa_rays_pose_backward + nsa_pose_grad_to_cam (g_cam[7]) and, with do_adam != 0, nsa_adam_step on the
 *   camera vector -- the same arithmetic in one block.  Reference: the lines those four entry points cite. */
int nsa_track_head(const float *uv, const float *K, const float *cam, uint32_t n, float *pose, float *rays_o,
                   float *rays_d, float *depth_scale, nsa_stream_t stream);
int nsa_track_tail(const float *uv, const float *K, float *cam, uint32_t n, const float *g_rays_o, const float *g_rays_d,
                   float *g_cam, int do_adam, float reduce_weight, float *exp_avg, float *exp_avg_sq, float *step, float lr,
                   float beta1, float beta2, float eps, uint32_t lr_step, float lr_gamma, const float *loss, float *best,
                   nsa_stream_t stream);
/* `best` (optional, 8 floats, with do_adam): the arg-min-loss camera of the frame -- best[0] = smallest loss[0] seen,
 * best[1..7] = the camera AFTER the step of that iteration (strict <).  replaces candidate_cam_tensor / current_min_loss
 * (code/training/volsdf_train.py:402-403,441-446).  Start a frame with best[0] = 1e10. */
/* Multi-GPU form: reduce_weight = this rank's ray count > 0 turns g_cam into the 9-float message
 * [w*g_cam(7), w*loss (slot 7 as left by nsa_l1_loss), w] that is summed over ranks with ONE all-reduce; the step is then
 * nsa_adam_step_scaled(cam, msg, msg + 8, ...) = Adam on msg[0..6] / msg[8]. */
/* The same iteration with the per-ray neighbours folded in (one ray chunk; what the graph-captured tracker runs):
 *   nsa_track_begin     = copy of the frame's pixel batch (uv_in [n,2], gt_in [n,3]) into the resident buffers uv / gt that the
 *                         captured sequence reads + nsa_track_head -- one launch in front of the graph replay;
 *   nsa_composite_track = nsa_composite_forward (rendered colour only) + nsa_l1_loss + nsa_composite_backward(g_rgb_values) in
 *                         one pass per ray; ray_loss[r] = sum_c |rgb_values[r,c] - gt[r,c]|, the cotangent is sign / (3 n_total);
 *                         g_grad is zero-filled (the tracking objective has no normal-map term)
 *                         (code/model/network.py:349-370, loss.py:57-65,131);
 *   nsa_track_finish    = nsa_rays_backward + nsa_track_tail, the loss (slot 7 of g_cam) formed as sum(ray_loss) / (3 n) from
 *                         fixed-order block partials by the last block to arrive: deterministic.  `workspace`:
 *                         nsa_track_finish_workspace(n) floats, zero-filled ONCE by the caller (the kernel leaves its ticket zero).
 *                         `best` compares g_cam[7]. */
int nsa_track_begin(const float *uv_in, const float *gt_in, float *uv, float *gt, const float *K, const float *cam, uint32_t n,
                    float *pose, float *rays_o, float *rays_d, float *depth_scale, nsa_stream_t stream);
/* nsa_track_begin and the iteration's sampler draws (nsa_draw with n_rand uniforms into t_rand, the n_extra picks of E into
 * extra_idx, no eikonal picks; R = n) in ONE launch: the draw's workgroups ride beside the ray lifting instead of being a graph
 * node of their own.  Same draws, bit for bit, as nsa_draw on the same state. */
int nsa_track_begin_draw(const float *uv_in, const float *gt_in, float *uv, float *gt, const float *K, const float *cam, uint32_t n,
                         float *pose, float *rays_o, float *rays_d, float *depth_scale, uint64_t *state, uint64_t n_rand,
                         float *t_rand, uint32_t E, uint32_t n_extra, uint32_t S, int32_t *extra_idx, nsa_stream_t stream);
int nsa_composite_track(const float *rays_o, const float *rays_d, const float *z_vals, const float *sdf, const float *rgb,
                        const float *voxels, uint32_t voxel_res, uint32_t R, uint32_t S, const float *gt, uint32_t n_total,
                        float *rgb_values, float *ray_loss, float *g_sdf, float *g_rgb, float *g_grad, nsa_stream_t stream);
int nsa_track_finish(const float *uv, const float *K, float *cam, uint32_t n, uint32_t S, const float *z_vals, const float *g_x,
                     const float *g_dir, const float *ray_loss, float *g_cam, int do_adam, float reduce_weight, float *exp_avg,
                     float *exp_avg_sq, float *step, float lr, float beta1, float beta2, float eps, uint32_t lr_step,
                     float lr_gamma, float *best, float *workspace, nsa_stream_t stream);
uint64_t nsa_track_finish_workspace(uint32_t n);
int nsa_adam_step_scaled(float *param, const float *grad, const float *grad_div, float *exp_avg, float *exp_avg_sq,
                         float *step, uint32_t n, float lr, float beta1, float beta2, float eps, uint32_t lr_step,
                         float lr_gamma, const float *loss, float *best, nsa_stream_t stream);
/* (loss, best: as for nsa_track_tail, best[1..n]; the loss compared is loss[0] / grad_div[0]) */

/* ---- Section 4: mapping-iteration tail ------------------------------------------------------------------------ */

/* keys[i] = 30-bit Morton code of point i in a 1024^3 lattice over [-1,1]^3; argsort(keys) is a spatially coherent
 * launch order for nsa_points_t.order (new -- the reference processes points in ray order). */
int nsa_morton_keys(const nsa_points_t *pts, int32_t *keys, nsa_stream_t stream);

/* order[] = stable argsort of the top `key_bits` (1..30) bits of those keys: the launch order itself, by an in-library LSD radix
 * sort (8-bit digits, two launches per pass, deterministic).  workspace: nsa_morton_order_workspace(P) 4-byte words.
 * (new, as nsa_morton_keys; replaces keys -> torch.sort -> indices.) */
int nsa_morton_order(const nsa_points_t *pts, int32_t *order, uint32_t *workspace, uint32_t key_bits, nsa_stream_t stream);
uint64_t nsa_morton_order_workspace(uint32_t P);

/* voxels[floor((x+1)/2*res)] += 1 for every sample with all |x_d| <= 0.99 (voxels: [res,res,res] fp32, x-major).
 * replaces SLAMNetwork.update_voxels (code/model/network.py:62-76). */
int nsa_update_voxels(const nsa_points_t *pts, float *voxels, uint32_t res, nsa_stream_t stream);

/* One torch.optim.Adam step (no weight decay / amsgrad) over n parameters in a single pass; `step` = this step's
 * number t >= 1 (bias corrections are computed on the host in double, like torch).  16-byte aligned pointers take the
 * 16-byte path; anything less (4-byte aligned at least) a one-element-per-thread form with the same arithmetic.
 * replaces self.optimizer.step() for one parameter tensor (code/training/volsdf_train.py:174, 420-424). */
int nsa_adam_table_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, uint64_t n, uint32_t step,
                        float lr, float beta1, float beta2, float eps, nsa_stream_t stream);

/* The same step with the gradient CONSUMED: every gradient element read is left zero (16-byte groups that were already zero are
 * not rewritten), so a persistent gradient buffer needs no zero fill before the next backward pass scatters into it.
 * replaces optimizer.step() + the next iteration's optimizer.zero_grad() / dense zero-initialised table gradient
 * (code/training/volsdf_train.py:547-576, code/hashencoder/hashgrid.py:117-118) for one table. */
int nsa_adam_table_step_clear(float *param, float *grad, float *exp_avg, float *exp_avg_sq, uint64_t n, uint32_t step,
                              float lr, float beta1, float beta2, float eps, nsa_stream_t stream);

/* The step of a parameter that received NO gradient this iteration, as the reference's environment takes it: under torch 1.11
 * (env_yamls/nicer-slam.yaml:62) optimizer.zero_grad() (code/training/volsdf_train.py:547) leaves zero TENSORS, so Adam still decays both
 * moments and moves the parameter along its momentum (the fine table during stage "coarse", the colour table during color_stage "base",
 * :550-555); torch >= 2.0 sets .grad = None and skips such a parameter.  Arithmetic of nsa_adam_table_step with grad == 0, no gradient
 * read (3 reads + 3 writes per element).  replaces zero_grad() + optimizer.step() for one un-touched tensor (volsdf_train.py:547,576). */
int nsa_adam_table_step_zero_grad(float *param, float *exp_avg, float *exp_avg_sq, uint64_t n, uint32_t step, float lr, float beta1,
                                  float beta2, float eps, nsa_stream_t stream);

/* The same Adam step over up to 24 SMALL tensors (n <= 2^24 each) in one launch: the weight_v / weight_g / bias tensors of the
 * trained MLPs.  Per-tensor lr and step count; beta1, beta2, eps shared.  4-byte aligned pointers suffice. */
typedef struct nsa_adam_seg {
    float *param;
    const float *grad;
    float *exp_avg, *exp_avg_sq;
    uint32_t n, step;
    float lr;
} nsa_adam_seg_t;
int nsa_adam_multi_step(const nsa_adam_seg_t *segs, uint32_t count, float beta1, float beta2, float eps, nsa_stream_t stream);

/* p[0..n) = 0 (p 16-byte aligned): the zero fill of a table-gradient buffer as a library launch, issued on the launch stream right
 * before the first backward kernel of a pass scatters into the buffer (nicer_slam_amd/fused/tablegrad.py).  replaces optimizer.zero_grad() +
 * the zero-initialised dense gradient of code/hashencoder/hashgrid.py:117-118. */
int nsa_fill_zero(float *p, uint64_t n, nsa_stream_t stream);

/* Weight-normed Linear layers -> the flat effective parameter vector the packed blocks and the MAP kernels' gradients use:
 *   flat = [W_0 (rows x cols, row-major), b_0, W_1, b_1, .., 0],   W_l[r,:] = weight_v_l[r,:] * weight_g_l[r] / ||weight_v_l[r,:]||,
 * norms[row] = ||weight_v_l[r,:]|| for the backward (rows of all layers, in order); n_layers <= 8.
 * replaces nn.utils.weight_norm's per-layer recomputation (torch._weight_norm, dim 0) of code/model/base_networks.py:137-141,
 * 376-379 + the reshape / cat that followed it here. */
typedef struct nsa_wn_layer {
    const float *weight_v;   /* [rows, cols] */
    const float *weight_g;   /* [rows]       */
    const float *bias;       /* [rows]       */
    uint32_t rows, cols;
} nsa_wn_layer_t;
int nsa_weight_norm_flat(const nsa_wn_layer_t *layers, uint32_t n_layers, float *flat, float *norms, nsa_stream_t stream);
/* Its backward: g_flat (layout of flat) -> g_params = per layer [g_weight_v (rows x cols) | g_weight_g (rows) | g_bias (rows)]. */
int nsa_weight_norm_flat_backward(const nsa_wn_layer_t *layers, uint32_t n_layers, const float *norms, const float *g_flat,
                                  float *g_params, nsa_stream_t stream);

/* One extra emission row for nsa_emit_gemm: dst[p] = src[order ? order[p] : p] (src NULL: `fill`) for p < P, 0 for P <= p < n. */
int nsa_emit_row(float *dst, const float *src, const int32_t *order, uint32_t P, uint64_t n, float fill, nsa_stream_t stream);

/* Packed MLP parameter block (MFMA fragment order) from the flat effective parameters in one launch:
 *   out[o] = word order[o] of concat( split(flat[a_index]) as [group][piece 0/1/2][lane][4 words of 2 halfwords], flat[v_index] )
 * (the pieces: nsa_operand_form)
 * a_index: n_a = groups * 64 * 8 indices into `flat` (the 8 fp32 weights lane l supplies to one MFMA k-group; every weight is
 * split exactly into three round-to-nearest bf16 pieces), v_index: n_v indices of per-feature values kept in fp32, order:
 * n_out <= n_a * 3 / 2 + n_v word indices into that concatenation.  The index arrays are the host-built layout tables of the
 * packed blocks (nicer_slam_amd/fused/pack.py; layout: csrc/mlp_common.hpp, csrc/mlp16.hpp).  New -- the reference has no
 * packed weights; replaces the per-layer weight_norm'ed nn.Linear weights of code/model/base_networks.py:127-149 as kernel input. */
/* How this library's fp32 path feeds the matrix cores, i.e. what nsa_pack_blocks writes into a fragment triple: 3 = three exact bf16
 * pieces of the weight (hi / mid / lo), 2 = two fp16 pieces of 512 w (round-to-nearest twice) + the bf16 round-to-nearest value for
 * the bf16-operand kernels.  A build-time choice (csrc/mlp_common.hpp::NSA_FORM); host code that builds packs itself (the
 * differentiable fallback of fused/pack.py) asks.  New -- no reference counterpart. */
int nsa_operand_form(void);

int nsa_pack_blocks(const float *flat, const int64_t *a_index, uint64_t n_a, const int64_t *v_index, uint64_t n_v,
                    const int64_t *order, uint64_t n_out, float *out, nsa_stream_t stream);

/* MLP weight gradients from the emission rows of the *_backward_params kernels: emit is [rows][ld] fp32 (column = point,
 * ld a multiple of 4096, columns past the last point zero).
 *   out[m][n] = sum_j sum_p emit[a_rows[j] + m][p] * emit[b_rows[j] + n][p],   j < pairs (1 or 2), m < M <= 64, n < N <= 192
 * and, with row_sums, out[m][N] = sum_p emit[a_rows[0] + m][p] (the bias gradient); out is [M][N + row_sums], fp32-faithful
 * products, deterministic (per-chunk partials in `workspace`, nsa_emit_gemm_workspace() floats, added in chunk order).
 * replaces torch autograd's weight / bias gradients of the Linear layers (code/model/base_networks.py:195-221, 333-395). */
int nsa_emit_gemm(const float *emit, uint64_t ld, uint32_t pairs, const uint32_t *a_rows, const uint32_t *b_rows, uint32_t M,
                  uint32_t N, int row_sums, float *out, float *workspace, nsa_stream_t stream);
uint64_t nsa_emit_gemm_workspace(uint64_t ld, uint32_t M, uint32_t N, int row_sums);

/* The per-ray terms of SLAMLoss (code/model/loss.py:113-233: rgb L1, eikonal, smooth, scale-and-shift-invariant monocular
 * depth with its alpha = 0.5 first-difference regulariser (code/utils/MiDaS.py:6-143), gt-depth L1, normal L1 + cos) and the
 * gradient of their WEIGHTED sum w.r.t. the model outputs, in three launches.  A weight of 0 skips a term (its value is then 0,
 * like the reference's 0.0); the flow and patch-warp terms are Section 5.  R = bs * n rays, image-major. */
typedef struct nsa_loss {
    uint32_t bs, n;               /* images, rays per image                                                             */
    uint32_t S;                   /* samples per ray of `sdf`                                                           */
    uint32_t E;                   /* eikonal points (0: no eikonal / smooth term)                                       */
    const float *rgb, *rgb_gt;    /* [R,3] rgb_values, ground_truth['rgb']                                              */
    const float *depth;           /* [R]   depth_values                                                                 */
    const float *depth_mono;      /* [R]   ground_truth['depth'] (monocular; the term aligns to 50 * it + 0.5)           */
    const float *depth_real;      /* [R]   target of the gt-depth L1 term (gt_depth, or depth * assign_scale on frame 0) */
    const float *depth_real_mask; /* [R]   the gt-depth term covers rays with this > 0 (ground_truth['gt_depth'])        */
    const float *mask_gt;         /* [R]   ground_truth['mask'] (foreground where > 0.5 AND the ray's sdf changes sign)  */
    const float *sdf;             /* [R,S] */
    const float *normal, *normal_gt;             /* [R,3] normal_map, ground_truth['normal']                            */
    const float *grad_theta, *grad_theta_nei;    /* [E,3]; grad_theta_nei may be NULL (no smooth term)                  */
    float w_rgb, w_eik, w_smooth, w_depth, w_gtdepth, w_nl1, w_ncos;
    int depth_whole_image;        /* depth term over every ray instead of the foreground (Replica scan 4, loss.py:171-175) */
    float *g_rgb, *g_depth, *g_normal, *g_theta, *g_theta_nei;   /* out: d(weighted sum) / d(input), same shapes         */
    float *terms;                 /* out [8]: rgb, eikonal, smooth, depth, gt_depth, normal_l1, normal_cos (unweighted), sum */
} nsa_loss_t;
int nsa_slam_loss(const nsa_loss_t *in, float *workspace /* nsa_slam_loss_workspace() floats, 8-byte aligned */,
                  nsa_stream_t stream);
uint64_t nsa_slam_loss_workspace(uint32_t bs, uint32_t n, uint32_t E);

/* ---- Section 5: keyframe re-projection blocks of a mapping iteration (patch warp, flow) and their masked-L1 terms ---- */

/* Shared description of a mapping batch: b keyframes x n sampled pixels.  `images` / `depths` are the resident full frames
 * ([frames,H,W,3] / [frames,H,W] fp32, pixel (y,x) at y*W+x -- the reference's ground_truth['full_rgb'] / ['full_depth'],
 * code/datasets/scene_dataset.py:248-257); batch entry i uses frame frame_index[i] of that store (NULL: frame i), so a
 * batch needs no per-iteration stacking copy of the frames. */
typedef struct nsa_warp {
    uint32_t b, n;              /* keyframes in the batch, sampled pixels per keyframe                                  */
    uint32_t H, W;              /* image size                                                                           */
    const float *uv;            /* [b,n,2] pixel coordinates                                                            */
    const float *pose;          /* [b,4,4] camera-to-world                                                              */
    const float *w2c;           /* [b,4,4] its inverse (the caller forms it with torch.linalg.inv like network.py:157,191) */
    const float *K;             /* [b,4,4] intrinsics                                                                   */
    const float *depth;         /* [b,n]   rendered depth along the ray, BEFORE the depth_scale factor (network.py:147-150) */
    const float *images;        /* [frames,H,W,3]                                                                       */
    const float *depths;        /* [frames,H,W]   (patch > 1 only; may be NULL otherwise)                               */
    const int32_t *frame_index; /* [b] device, or NULL                                                                  */
} nsa_warp_t;

/* Patch warp, forward: for every (target t, source s, pixel i, patch cell c) the reference's warp_output[patch] tensors
 *   sampled[t,s,i,c,3] = bilinear sample (zeros padding, align_corners=True) of image t at the projection of the lifted cell,
 *   gt_rgb [t,s,i,c,3] = image s at the cell's own pixel (ones outside the image), replicated over t,
 *   mask   [t,s,i,c]   = projection strictly inside image t and in front of it  &  cell inside image s  &  (patch > 1) flat[s,i],
 *   flat   [s,i]       = biased variance of the patch's ground-truth depths < 0.01 (patch > 1; else untouched, may be NULL).
 * Cell order c = ix * patch + iy with offsets (ix - patch/2, iy - patch/2) on (u,v) (general.py:139-144); patch must be odd.
 * replaces code/model/network.py:167-279 (one patch size per call) + uv2patch (code/utils/general.py:129-145). */
int nsa_patch_warp_forward(const nsa_warp_t *in, uint32_t patch, float *sampled, uint8_t *mask, float *gt_rgb, uint8_t *flat,
                           nsa_stream_t stream);

/* Backward of the above: g_sampled[t,s,i,c,3] -> g_depth[b,n] (overwritten); when g_pose and g_w2c are non-NULL also the
 * gradients of the source poses (through ray origin and direction) and of the target world-to-camera matrices, both [b,4,4]
 * (overwritten, bottom row zero) -- bundle adjustment (volsdf_train.py:521-528).  Deterministic (fixed-order sums).
 * workspace: nsa_patch_warp_workspace() floats (may be NULL when that is 0). */
int nsa_patch_warp_backward(const nsa_warp_t *in, uint32_t patch, const float *g_sampled, float *g_depth, float *g_pose,
                            float *g_w2c, float *workspace, nsa_stream_t stream);
uint64_t nsa_patch_warp_workspace(uint32_t b, uint32_t n, uint32_t patch, int want_pose);

/* Flow: flow[e,i,2] = projection into frame idjj[e] of the rendered point of pixel i of frame idii[e], minus that pixel.
 * idii / idjj: [ne] int64 on the device (the reference's `edges`, volsdf_train.py:312-324).  replaces network.py:153-165.
 * (`images`, `depths` of nsa_warp_t are not used.) */
int nsa_flow_forward(const nsa_warp_t *in, const int64_t *idii, const int64_t *idjj, uint32_t ne, float *flow,
                     nsa_stream_t stream);
int nsa_flow_backward(const nsa_warp_t *in, const int64_t *idii, const int64_t *idjj, uint32_t ne, const float *g_flow,
                      float *g_depth, float *g_pose, float *g_w2c, float *workspace, nsa_stream_t stream);
uint64_t nsa_flow_workspace(uint32_t b, uint32_t n, uint32_t ne, int want_pose);

/* loss[0] = mean over the selected items and their `channels` values of |pred - target| (NaN for an empty selection, like
 * torch); g_pred (optional, [items,channels]) = its gradient, 0 outside the mask.  mask: [items] bytes or NULL (all).
 * replaces `(sampled[mask] - gt[mask]).abs().mean()` (code/model/loss.py:136-142) and the flow L1 on flow_mask (:106-111).
 * workspace: nsa_masked_l1_workspace() floats, 8-byte aligned.  Deterministic. */
int nsa_masked_l1(const float *pred, const float *target, const uint8_t *mask, uint64_t items, uint32_t channels, float *loss,
                  float *g_pred, float *workspace, nsa_stream_t stream);
uint64_t nsa_masked_l1_workspace(uint64_t items);

/* ---- Section 6: per-iteration input batch from frames resident in HBM ------------------------------------------------ */

/* One field of the frame stores: store [capacity, pixels, channels] fp32, out [b, n, channels]. */
typedef struct nsa_feed_field {
    const float *store;
    float *out;
    uint32_t channels; /* 1..16 */
} nsa_feed_field_t;

/* out_f[i,k,:] = store_f[slots[i], sel[k], :] for every field (at most 8), uv[i,k] = (sel[k] % width, sel[k] / width) (uv may be
 * NULL) in one launch.  slots: [b] int32 store slots of the batch's frames, sel: [n] int64 pixel indices (an index outside
 * [0, pixels) yields NaN rows).  replaces SLAMDataset.__getitem__'s `[self.sampling_idx, :]` of every image + collate_fn
 * (code/datasets/scene_dataset.py:214-275) for frames kept on the device (nicer_slam_amd/feed.py). */
int nsa_feed_gather(const nsa_feed_field_t *fields, uint32_t n_fields, const int32_t *slots, uint32_t b, const int64_t *sel,
                    uint32_t n, uint64_t pixels, uint32_t width, float *uv, nsa_stream_t stream);

/* Up to 8 float segments (dst[i][0..n) = src[i][0..n), device pointers) copied in ONE launch: the per-call inputs of a cached graph
 * (pose, pixel batch, intrinsics -- the `.cuda()` / copy of each model input in the reference's loop, volsdf_train.py:411-416). */
typedef struct nsa_copy_seg {
    float *dst;
    const float *src;
    uint32_t n; /* floats */
} nsa_copy_seg_t;
int nsa_copy_segments(const nsa_copy_seg_t *segs, uint32_t n_segs, nsa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NICER_SLAM_AMD_H */
