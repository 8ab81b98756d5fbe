"""Pose parametrisation on the pose-gradient path (reference code/utils/general.py:52-100)
and the config-string class resolver that makes ``train.model_class`` a plug-in point (general.py:153-159)."""
import importlib

import torch


def quad2rotation(quad):
    """(w, x, y, z) quaternion batch -> rotation matrices with two_s = 2/|q|^2 (general.py:52-76).
    Built with stack (differentiable, no in-place writes, stays on quad's device)."""
    qr, qi, qj, qk = quad[:, 0], quad[:, 1], quad[:, 2], quad[:, 3]
    two_s = 2.0 / (quad * quad).sum(-1)
    r0 = torch.stack([-two_s * (qj * qj + qk * qk) + 1, two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr)], -1)
    r1 = torch.stack([two_s * (qi * qj + qk * qr), -two_s * (qi ** 2 + qk ** 2) + 1, two_s * (qj * qk - qi * qr)], -1)
    r2 = torch.stack([two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), -two_s * (qi ** 2 + qj ** 2) + 1], -1)
    return torch.stack([r0, r1, r2], 1)


class _CamToPose(torch.autograd.Function):
    """cam[b,7] -> pose[b,4,4] and its backward as ONE kernel each (csrc/track_tail.hip: k_cam_to_pose / k_pose_grad_to_cam,
    C ABI nsa_cam_to_pose / nsa_pose_grad_to_cam) instead of the ~45 + ~90 element-wise launches torch makes of quad2rotation and
    its autograd graph -- per iteration of the reference's tracking loop that is a third of all launches."""

    @staticmethod
    def forward(ctx, cam):
        from .._native import lib, check
        cam = cam.contiguous()
        pose = torch.empty(cam.shape[0], 4, 4, device=cam.device, dtype=torch.float32)
        check(lib.nsa_cam_to_pose(cam.data_ptr(), cam.shape[0], pose.data_ptr(), torch.cuda.current_stream().cuda_stream))
        ctx.save_for_backward(cam)
        return pose

    @staticmethod
    def backward(ctx, g_pose):
        from .._native import lib, check
        (cam,) = ctx.saved_tensors
        g_pose = g_pose.contiguous()
        g_cam = torch.empty_like(cam)
        check(lib.nsa_pose_grad_to_cam(cam.data_ptr(), g_pose.data_ptr(), cam.shape[0], g_cam.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream))
        return g_cam


def get_camera_from_tensor(inputs):
    """7-vector (quaternion wxyz, translation) -> 4x4 camera-to-world (general.py:79-100).  Device float32 tensors go through the
    HIP kernel pair (forward + backward); anything else through the torch restatement below."""
    single = inputs.dim() == 1
    if single:
        inputs = inputs.unsqueeze(0)
    if inputs.is_cuda and inputs.dtype == torch.float32 and inputs.shape[-1] == 7:
        RT = _CamToPose.apply(inputs)
        return RT[0] if single else RT
    return camera_from_tensor_torch(inputs[0] if single else inputs)


def camera_from_tensor_torch(inputs):
    """The same map with torch ops in the reference's operation order (general.py:52-100), on any device.  The kernel agrees with it
    to the last ulps; where a test needs the pose bit-for-bit as the reference's ops build it (a ray's far sample sits exactly on
    the cube face), it asks for this one."""
    single = inputs.dim() == 1
    if single:
        inputs = inputs.unsqueeze(0)
    R = quad2rotation(inputs[:, :4])
    top = torch.cat([R, inputs[:, 4:, None]], 2)
    bottom = top.new_zeros(top.shape[0], 1, 4)      # built on the device: no H2D copy (hipGraph-capture safe)
    bottom[:, :, 3] = 1.0
    RT = torch.cat([top, bottom], 1)
    return RT[0] if single else RT


def get_tensor_from_camera(RT, Tquad=False):
    """4x4 (or 3x4) camera matrix -> 7-vector, without the reference's Blender ``mathutils`` dependency
    (general.py:103-126).  Shepperd's method; returns the w >= 0 representative like Matrix.to_quaternion()."""
    dev = RT.device if isinstance(RT, torch.Tensor) else "cpu"
    M = torch.as_tensor(RT, dtype=torch.float64).detach().cpu()
    R, T = M[:3, :3], M[:3, 3]
    tr = R.trace()
    if tr > 0:
        s = torch.sqrt(tr + 1.0) * 2
        q = torch.stack([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(torch.argmax(torch.diagonal(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = torch.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = torch.zeros(4, dtype=torch.float64)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    if q[0] < 0:
        q = -q
    out = torch.cat([T, q]) if Tquad else torch.cat([q, T])
    return out.float().to(dev)


def get_class(kls):
    """'pkg.module.Class' -> class (general.py:153-159)."""
    module, _, name = kls.rpartition(".")
    return getattr(importlib.import_module(module), name)


def index_to_1d(idx, res):
    """[N,3] integer voxel index -> flat index, x-major (general.py index_to_1d)."""
    return idx[:, 0] * res * res + idx[:, 1] * res + idx[:, 2]
