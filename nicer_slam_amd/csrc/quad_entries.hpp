// Prototypes of the quad-tiling entry points (csrc/render_sdfnet4.hip, csrc/render_sampler4.hip): internal, reached through the
// public entry points of render_sdfnet.hip / render_sampler.hip when nsa_grid_t.tile == 16.  NSA_ENTRY gives the bf16-operand
// build its own set (see bf16_entries.hpp).
#pragma once
#include "../../include/nicer_slam_amd.h"
#ifndef NSA_ENTRY
#define NSA_ENTRY(x) x
#endif
extern "C" {
int NSA_ENTRY(nsa_sdfnet4_forward)(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, int accumulate, float* sdf, float* grad, float* feat_hl, nsa_stream_t stream);
int NSA_ENTRY(nsa_sdfnet4_backward)(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, const float* g_sdf, const float* g_feat_hl, const float* g_grad, int accumulate, float* g_x, float* g_table, float* emit, uint32_t emit_ld, nsa_stream_t stream);
int NSA_ENTRY(nsa_sdfnet4_forward_pair)(const nsa_points_t* pts, const nsa_grid_t* coarse, const nsa_grid_t* fine, const float* packed_coarse, const float* packed_fine, float* sdf, float* grad, float* feat_hl, nsa_stream_t stream);
int NSA_ENTRY(nsa_sdfnet4_emit_rows)(uint32_t n_hidden);
int NSA_ENTRY(nsa_sampler4_sdf)(const float* rays_o, const float* rays_d, uint32_t R, uint32_t E, const float* t_lin, const float* t_rand, float near, float bound, float far_cap, const nsa_grid_t* coarse, const nsa_grid_t* fine, const float* packed_coarse, const float* packed_fine, float* z, float* sdf, float* far, nsa_stream_t stream);
int NSA_ENTRY(nsa_sdf4_points)(const float* points, uint64_t N, const nsa_grid_t* coarse, const nsa_grid_t* fine, const float* packed_coarse, const float* packed_fine, float* sdf, nsa_stream_t stream);
}
