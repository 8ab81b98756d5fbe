"""Pixel -> ray lifting on the hot path (reference code/utils/rend_util.py:68-93,107-129)."""
import torch


def lift(x, y, z, intrinsics):
    """Back-project pixel (x, y) at depth z through K (with skew) -> homogeneous camera point (:107-129)."""
    fx, fy = intrinsics[:, 0, 0].unsqueeze(-1), intrinsics[:, 1, 1].unsqueeze(-1)
    cx, cy = intrinsics[:, 0, 2].unsqueeze(-1), intrinsics[:, 1, 2].unsqueeze(-1)
    sk = intrinsics[:, 0, 1].unsqueeze(-1)
    x_lift = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    y_lift = (y - cy) / fy * z
    return torch.stack((x_lift, y_lift, z, torch.ones_like(z)), dim=-1)


def get_camera_params(uv, pose, intrinsics):
    """uv[b,n,2], pose[b,4,4] (c2w), K[b,4,4] -> (ray_dirs[b,n,3], cam_loc[b,3]).

    NB ray directions are divided by their SQUARED norm (rend_util.py:92); all z values downstream are in
    those units.  Only the 4x4 pose branch is implemented (the 7-vector branch of :69-74 is unused by the
    training loop, which converts with get_camera_from_tensor first)."""
    if pose.shape[1] == 7:
        raise NotImplementedError("pass a 4x4 pose (utils.general.get_camera_from_tensor)")
    cam_loc = pose[:, :3, 3]
    x_cam, y_cam = uv[:, :, 0], uv[:, :, 1]
    pts = lift(x_cam, y_cam, torch.ones_like(x_cam), intrinsics.to(uv.device)).permute(0, 2, 1)
    world = torch.bmm(pose, pts).permute(0, 2, 1)[:, :, :3]
    ray_dirs = world - cam_loc[:, None, :]
    ray_dirs = ray_dirs / (ray_dirs * ray_dirs).sum(-1, keepdim=True)
    return ray_dirs, cam_loc
