// Prototypes of the bf16-operand variants of the MLP entry points (defined in csrc/*_bf16.hip).
#pragma once
#include "../../include/nicer_slam_amd.h"
extern "C" {
int nsa_sdfnet_forward_bf16(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, int accumulate, float* sdf, float* grad, float* feat_hl, nsa_stream_t stream);
int nsa_sdfnet_forward_pair_bf16(const nsa_points_t* pts, const nsa_grid_t* coarse, const nsa_grid_t* fine, const float* packed_coarse, const float* packed_fine, float* sdf, float* grad, float* feat_hl, nsa_stream_t stream);
int nsa_sdfnet_backward_bf16(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, const float* g_sdf, const float* g_feat_hl, const float* g_grad, int accumulate, float* g_x, nsa_stream_t stream);
int nsa_sdfnet_backward_params_bf16(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, const float* g_sdf, const float* g_feat_hl, const float* g_grad, int accumulate, float* g_x, float* g_table, float* emit, uint32_t emit_ld, nsa_stream_t stream);
int nsa_colour_forward_bf16(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, const float* grad, const float* feat_hl, float* rgb, float* save, nsa_stream_t stream);
int nsa_colour_forward_track_bf16(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, const float* grad, const float* feat_hl, float* rgb, float* save, const float* sdf, const float* voxels, uint32_t voxel_res, const float* gt, uint32_t n_total, float* rgb_values, float* ray_loss, float* g_sdf, float* g_rgb, float* g_grad, nsa_stream_t stream);
int nsa_colour_backward_bf16(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, const float* grad, const float* feat_hl, const float* save, const float* g_rgb, int grid_grad, float* g_feat_hl, float* g_grad, float* g_x, float* g_dir, nsa_stream_t stream);
int nsa_colour_coarse_backward_bf16(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, const float* grad, const float* feat_hl, const float* save, const float* g_rgb, int grid_grad, float* g_feat_hl, float* g_grad, float* g_x, float* g_dir, const nsa_grid_t* coarse, const float* packed_coarse, const float* g_sdf, nsa_stream_t stream);
int nsa_colour_backward_params_bf16(const nsa_points_t* pts, const nsa_grid_t* grid, const float* packed, const float* grad, const float* feat_hl, const float* save, const float* g_rgb, int grid_grad, float* g_feat_hl, float* g_grad, float* g_x, float* g_dir, float* g_table, float* emit, uint32_t emit_ld, nsa_stream_t stream);
int nsa_sampler_sdf_bf16(const float* rays_o, const float* rays_d, uint32_t R, uint32_t E, const float* t_lin, const float* t_rand, float near, float bound, float far_cap, const nsa_grid_t* coarse, const nsa_grid_t* fine, const float* packed_coarse, const float* packed_fine, float* z, float* sdf, float* far, nsa_stream_t stream);
int nsa_sdf_points_bf16(const float* points, uint64_t N, const nsa_grid_t* coarse, const nsa_grid_t* fine, const float* packed_coarse, const float* packed_fine, float* sdf, nsa_stream_t stream);
}
