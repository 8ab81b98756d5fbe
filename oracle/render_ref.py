"""oracle/render_ref.py -- TEST INFRASTRUCTURE ONLY (never on the product path).

Functional, pure-PyTorch **CPU** restatement of the reference's per-frame neural rendering
core, with the hash encoder provided by the C oracle
(``oracle/hashenc_oracle.c``).  It is the checker for the HIP path (tests/), the smoke
check, and the timed ``cpu_baseline`` ("port") in bench.py.  Every function cites the
reference lines it restates; parameters are addressed by the reference's ``state_dict`` names.

Parity pin: ``tests/golden/*.npz`` were produced by running the reference's own Python
(imported from /root/reference in the build container, ``tests/golden/make_golden.py``) and
``tests/test_oracle_golden.py`` holds this file to them.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import hashenc

_BACKEND = hashenc.OracleBackend()


# ----------------------------------------------------------------------------- grid layout
@dataclass
class GridSpec:
    """Level/offset layout of one multi-resolution grid (code/hashencoder/hashgrid.py:141-178)."""
    num_levels: int
    level_dim: int
    base_resolution: int
    per_level_scale: float
    log2_hashmap_size: int
    offsets: torch.Tensor  # int32 [L+1]
    input_dim: int = 3

    @property
    def n_rows(self):
        return int(self.offsets[-1])

    @property
    def out_dim(self):
        return self.num_levels * self.level_dim


def make_grid_spec(num_levels, level_dim, base_resolution, desired_resolution, log2_hashmap_size,
                   per_level_scale=2.0, input_dim=3):
    # hashgrid.py:144-146: desired_resolution overrides per_level_scale
    if desired_resolution is not None:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    cap = 2 ** log2_hashmap_size
    offs, total = [], 0
    for i in range(num_levels):  # hashgrid.py:163-170
        res = int(np.ceil(base_resolution * per_level_scale ** i))
        offs.append(total)
        total += min(cap, res ** input_dim)
    offs.append(total)
    return GridSpec(num_levels, level_dim, base_resolution, float(per_level_scale), log2_hashmap_size,
                    torch.from_numpy(np.array(offs, dtype=np.int32)), input_dim)


# -------------------------------------------------------------- hash encode autograd pair
class _EncodeBwd(torch.autograd.Function):
    """First backward as a differentiable node (hashgrid.py:79-134)."""

    @staticmethod
    def forward(ctx, grad, inputs, emb, offsets, dims, calc_gi, dy_dx):
        B, D, C, L, S, H = dims
        g_in = torch.zeros_like(inputs)
        g_emb = torch.zeros_like(emb)
        _BACKEND.hash_encode_backward(grad, inputs, emb, offsets, g_emb, B, D, C, L, S, H, calc_gi,
                                      dy_dx, g_in)
        ctx.save_for_backward(grad, inputs, emb, offsets, dy_dx)
        ctx.dims, ctx.calc_gi = dims, calc_gi
        return g_in, g_emb

    @staticmethod
    def backward(ctx, gg_in, gg_emb_unused):
        grad, inputs, emb, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H = ctx.dims
        grad_grad = torch.zeros_like(grad)
        g2_emb = torch.zeros_like(emb)
        _BACKEND.hash_encode_second_backward(grad, inputs, emb, offsets, B, D, C, L, S, H,
                                             ctx.calc_gi, dy_dx, gg_in.contiguous(), grad_grad, g2_emb)
        # hashgrid.py:134 -- no derivative w.r.t. the inputs is returned (term dropped)
        return grad_grad, None, g2_emb, None, None, None, None


class _Encode(torch.autograd.Function):
    """hashgrid.py:13-69"""

    @staticmethod
    def forward(ctx, inputs, emb, offsets, per_level_scale, base_resolution, calc_gi):
        inputs = inputs.contiguous()
        emb = emb.contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = emb.shape[1]
        S = np.log2(per_level_scale)
        H = base_resolution
        out = torch.empty(L, B, C, dtype=inputs.dtype)
        dy_dx = torch.empty(B, L * D * C, dtype=inputs.dtype) if calc_gi else torch.empty(1, dtype=inputs.dtype)
        _BACKEND.hash_encode_forward(inputs, emb, offsets, out, B, D, C, L, S, H, calc_gi, dy_dx)
        ctx.save_for_backward(inputs, emb, offsets, dy_dx)
        ctx.dims, ctx.calc_gi = (B, D, C, L, S, H), calc_gi
        return out.permute(1, 0, 2).reshape(B, L * C)

    @staticmethod
    def backward(ctx, grad):
        inputs, emb, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H = ctx.dims
        grad = grad.view(B, L, C).permute(1, 0, 2).contiguous()
        g_in, g_emb = _EncodeBwd.apply(grad, inputs, emb, offsets, ctx.dims, ctx.calc_gi, dy_dx)
        return (g_in if ctx.calc_gi else None), g_emb, None, None, None, None


def grid_features(x, emb, spec: GridSpec, size=1.0):
    """HashEncoder.forward (hashgrid.py:199-215): map [-size,size] -> [0,1] and encode."""
    u = (x + size) / (2 * size)
    shape = list(u.shape[:-1])
    u = u.view(-1, spec.input_dim)
    out = _Encode.apply(u, emb, spec.offsets, spec.per_level_scale, spec.base_resolution,
                        u.requires_grad)
    return out.view(shape + [spec.out_dim])


# ------------------------------------------------------------------------------ small nets
def positional_encoding(x, n_freq):
    """embedder.py:5-37,71-88 ('nerf' type: include input, log-sampled bands, sin then cos)."""
    bands = 2.0 ** torch.linspace(0.0, n_freq - 1, n_freq)
    parts = [x]
    for f in bands:
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return torch.cat(parts, -1)


# ---- optional bf16-OPERAND GEMM emulation (BASELINE configs[2] "bf16 MLP", configs[4] "bf16 + fp32 SDF head") ----------------
# The reference has no such mode (fp32 only; hashgrid.py:15's autocast is never enabled): this restates what the product's
# `mlp_precision = "bf16" | "bf16_colour"` kernels compute (nicer_slam_amd/csrc/mlp_common.hpp, NSA_PIECES == 1), so that those
# modes have a checker that is independent of the HIP code:
#   * every matrix-core GEMM takes BOTH operands rounded to bfloat16, round-to-nearest-even (weights: the bf16 round-to-nearest piece of the
#     packed split, fused/pack.py::split_bf16x3; activations / cotangents / tangents: v_cvt_pk_bf16_f32), products exact,
#     accumulation in fp32 starting from the fp32 bias;
#   * this holds for the forward GEMMs, for the reverse pass that builds grad sdf (cotangent operand rounded) and for the GEMMs of
#     the backward kernels (second-order sweeps included): _LinQ's backward is itself a _LinQ on the rounded cotangent;
#   * what the kernels compute on the vector ALU stays fp32: the sdf row of an SDF network's last layer (sdf_net4.hpp /
#     render_sdfnet4.hip: a VALU dot with the fp32 row), the colour network's 64 -> 3 output layer (render_colour.hip::colour_mlp),
#     encoders, activations, compositing.
def _rne_bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


class _RoundSTE(torch.autograd.Function):
    """x -> bf16(x) (round to nearest even), derivative 1: the rounding of a GEMM operand is not part of the differentiated graph."""

    @staticmethod
    def forward(ctx, x):
        return _rne_bf16(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _LinQ(torch.autograd.Function):
    """y = xr @ wr^T on ALREADY ROUNDED operands; every derivative GEMM rounds the vector it multiplies, recursively."""

    @staticmethod
    def forward(ctx, xr, wr):
        ctx.save_for_backward(xr, wr)
        return xr @ wr.t()

    @staticmethod
    def backward(ctx, g):
        xr, wr = ctx.saved_tensors
        gr = _RoundSTE.apply(g)
        gx = _LinQ.apply(gr, wr.t().contiguous()) if ctx.needs_input_grad[0] else None
        gw = _LinQ.apply(gr.t().contiguous(), xr.t().contiguous()) if ctx.needs_input_grad[1] else None
        return gx, gw


def q_linear(x, w, b):
    return _LinQ.apply(_RoundSTE.apply(x), _RoundSTE.apply(w)) + b


def wn_linear(params, prefix, x, quant=False, exact_rows=0):
    """nn.utils.weight_norm(nn.Linear) forward (base_networks.py:148-149,325-326): W = g*v/|v|_row.
    quant: the bf16-operand emulation above; the first ``exact_rows`` output rows stay plain fp32 (vector-ALU rows)."""
    g, v, b = params[prefix + ".weight_g"], params[prefix + ".weight_v"], params[prefix + ".bias"]
    w = v * (g / v.norm(2, dim=1, keepdim=True))
    if not quant:
        return F.linear(x, w, b)
    if exact_rows >= w.shape[0]:
        return F.linear(x, w, b)
    if exact_rows == 0:
        return q_linear(x, w, b)
    return torch.cat((F.linear(x, w[:exact_rows], b[:exact_rows]), q_linear(x, w[exact_rows:], b[exact_rows:])), dim=-1)


@dataclass
class SdfNetSpec:
    grid: GridSpec
    n_linear: int            # number of lin layers
    multires: int = 6
    divide_factor: float = 1.0


@dataclass
class RenderConfig:
    coarse: SdfNetSpec
    fine: SdfNetSpec
    colour_grid: GridSpec
    colour_n_linear: int = 3
    multires_view: int = 4
    colour_divide_factor: float = 1.0
    feature_vector_size: int = 64
    scene_bounding_sphere: float = 1.0
    near: float = 0.0
    n_samples: int = 64
    n_samples_eval: int = 640
    n_samples_extra: int = 32
    voxel_res: int = 64
    white_bkgd: bool = False
    mlp_precision: str = "fp32"     # "fp32" (the reference) | "bf16" (every MLP) | "bf16_colour" (colour MLP only): see _LinQ


def sdf_net_forward(params, prefix, spec: SdfNetSpec, x, quant=False):
    """ImplicitNetworkGrid.forward (base_networks.py:155-186): hash(x/df) ++ PE -> softplus MLP.
    quant: bf16-operand GEMMs (the sdf row of the last layer stays fp32, see _LinQ)."""
    feat = grid_features(x / spec.divide_factor, params[prefix + ".encoding.embeddings"], spec.grid)
    h = torch.cat((positional_encoding(x, spec.multires), feat), dim=-1)
    for l in range(spec.n_linear):
        h = wn_linear(params, f"{prefix}.lin{l}", h, quant, exact_rows=1 if l == spec.n_linear - 1 else 0)
        if l < spec.n_linear - 1:
            h = F.softplus(h, beta=100)
    return h


def _net_outputs(params, prefix, spec, x, quant=False):
    """ImplicitNetworkGrid.get_outputs (base_networks.py:208-221)."""
    x.requires_grad_(True)
    out = sdf_net_forward(params, prefix, spec, x, quant)
    sdf, feat = out[:, :1], out[:, 1:]
    (g,) = torch.autograd.grad(sdf, x, torch.ones_like(sdf), create_graph=True, retain_graph=True)
    return sdf, feat, g


def sdf_outputs(params, cfg: RenderConfig, x, stage="fine"):
    """ImplicitNetworkGrid_COMBINE.get_outputs (base_networks.py:34-40)."""
    q = cfg.mlp_precision == "bf16"
    c = _net_outputs(params, "implicit_network.coarse", cfg.coarse, x, q)
    if stage == "coarse":
        return c
    f = _net_outputs(params, "implicit_network.fine", cfg.fine, x, q)
    return c[0] + f[0], c[1] + f[1], c[2] + f[2]


def sdf_vals(params, cfg: RenderConfig, x, stage="fine"):
    """ImplicitNetworkGrid_COMBINE.get_sdf_vals (base_networks.py:27-32); the reference's extra
    coarse feature evaluation (:31) has no effect on the value and is not repeated."""
    q = cfg.mlp_precision == "bf16"
    s = sdf_net_forward(params, "implicit_network.coarse", cfg.coarse, x, q)[:, :1]
    if stage == "coarse":
        return s
    return s + sdf_net_forward(params, "implicit_network.fine", cfg.fine, x, q)[:, :1]


def sdf_gradient(params, cfg: RenderConfig, x, stage="fine"):
    """ImplicitNetworkGrid_COMBINE.gradient (base_networks.py:42-47,195-206)."""
    x.requires_grad_(True)

    q = cfg.mlp_precision == "bf16"

    def one(prefix, spec):
        y = sdf_net_forward(params, prefix, spec, x, q)[:, :1]
        return torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True, retain_graph=True)[0]

    g = one("implicit_network.coarse", cfg.coarse)
    if stage != "coarse":
        g = g + one("implicit_network.fine", cfg.fine)
    return g


def colour_net(params, cfg: RenderConfig, points, normals, view_dirs, feats, color_stage="highfreq"):
    """RenderingNetwork.forward, mode 'idr' with grid feature (base_networks.py:333-395)."""
    gf = grid_features(points / cfg.colour_divide_factor, params["rendering_network.encoding.embeddings"],
                       cfg.colour_grid)
    if color_stage == "base":
        gf = gf.detach()
    h = torch.cat([points, positional_encoding(view_dirs, cfg.multires_view), normals, feats, gf], dim=-1)
    q = cfg.mlp_precision in ("bf16", "bf16_colour")
    for l in range(cfg.colour_n_linear):
        h = wn_linear(params, f"rendering_network.lin{l}", h, q and l < cfg.colour_n_linear - 1)   # (64 -> 3: vector ALU, fp32)
        if l < cfg.colour_n_linear - 1:
            h = torch.relu(h)
    return torch.sigmoid(h)


def colour_relu_margin(params, cfg: RenderConfig, points, normals, view_dirs, feats):
    """min |pre-activation| over the colour MLP's ReLU units, per point (test helper): where it is ~1e-7 of the typical
    magnitude the unit's mask -- hence the reference's own gradient -- is decided by fp32 rounding noise."""
    with torch.no_grad():
        gf = grid_features(points / cfg.colour_divide_factor, params["rendering_network.encoding.embeddings"], cfg.colour_grid)
        h = torch.cat([points, positional_encoding(view_dirs, cfg.multires_view), normals, feats, gf], dim=-1)
        margin = torch.full((points.shape[0],), float("inf"))
        for l in range(cfg.colour_n_linear - 1):
            a = wn_linear(params, f"rendering_network.lin{l}", h)
            margin = torch.minimum(margin, a.abs().amin(dim=1))
            h = torch.relu(a)
    return margin


# --------------------------------------------------------------------------------- density
def beta_from_voxels(voxels, x, voxel_res):
    """GridPredefineDensity.func (density.py:41-60)."""
    oob = (x.abs() > 0.99).any(dim=1)
    idx = ((x + 1) / 2 * voxel_res).long()
    idx = torch.where(oob[:, None], torch.zeros_like(idx), idx)
    count = voxels[idx[:, 0], idx[:, 1], idx[:, 2]]
    count = torch.where(oob, torch.zeros_like(count), count)
    a, b, c, d = 0.01207724805, 0.0116544676, 0.0023639156, 5.37538
    return (a * torch.exp(-b * 0.0001 * count * d) + c).unsqueeze(-1)


def density(sdf, x, voxels, voxel_res):
    """GridPredefineDensity.density_func (density.py:37-39)."""
    beta = beta_from_voxels(voxels, x, voxel_res)
    alpha = 1 / beta
    return alpha * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))


def volume_weights(z_vals, sdf, points_flat, voxels, voxel_res):
    """SLAMNetwork.volume_rendering (network.py:349-370)."""
    sigma = density(sdf, points_flat, voxels, voxel_res).reshape(-1, z_vals.shape[1])
    dists = z_vals[:, 1:] - z_vals[:, :-1]
    dists = torch.cat([dists, torch.full((dists.shape[0], 1), 1e10)], -1)
    energy = dists * sigma
    shifted = torch.cat([torch.zeros(dists.shape[0], 1), energy[:, :-1]], dim=-1)
    alpha = -torch.exp(-energy) + 1
    trans = torch.exp(-torch.cumsum(shifted, dim=-1))
    return alpha * trans


def update_voxels(voxels, x, voxel_res):
    """SLAMNetwork.update_voxels (network.py:62-76) -> new counter tensor."""
    keep = ~(x.abs() > 0.99).any(dim=1)
    idx = ((x[keep] + 1) / 2 * voxel_res).long()
    flat = idx[:, 0] * voxel_res * voxel_res + idx[:, 1] * voxel_res + idx[:, 2]  # general.index_to_1d
    out = voxels.reshape(-1).clone()
    out.index_add_(0, flat, torch.ones_like(flat).float())
    return out.reshape(voxels.shape)


# ------------------------------------------------------------------------------------ rays
def camera_rays(uv, pose, intrinsics):
    """rend_util.get_camera_params + lift (rend_util.py:68-93,107-129), 4x4 pose branch.
    NB: directions are divided by their SQUARED norm (:92)."""
    cam_loc = pose[:, :3, 3]
    fx, fy = intrinsics[:, 0, 0].unsqueeze(-1), intrinsics[:, 1, 1].unsqueeze(-1)
    cx, cy = intrinsics[:, 0, 2].unsqueeze(-1), intrinsics[:, 1, 2].unsqueeze(-1)
    sk = intrinsics[:, 0, 1].unsqueeze(-1)
    x, y = uv[:, :, 0], uv[:, :, 1]
    z = torch.ones_like(x)
    x_l = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    y_l = (y - cy) / fy * z
    pts = torch.stack((x_l, y_l, z, torch.ones_like(z)), dim=-1).permute(0, 2, 1)
    world = torch.bmm(pose, pts).permute(0, 2, 1)[:, :, :3]
    d = world - cam_loc[:, None, :]
    d = d / (d * d).sum(-1, keepdim=True)
    return d, cam_loc


def quad2rotation(q):
    """general.py:52-76."""
    qr, qi, qj, qk = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    two_s = 2.0 / (q * q).sum(-1)
    rows = [
        torch.stack([-two_s * (qj * qj + qk * qk) + 1, two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr)], -1),
        torch.stack([two_s * (qi * qj + qk * qr), -two_s * (qi ** 2 + qk ** 2) + 1, two_s * (qj * qk - qi * qr)], -1),
        torch.stack([two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), -two_s * (qi ** 2 + qj ** 2) + 1], -1),
    ]
    return torch.stack(rows, 1)


def camera_from_tensor(t):
    """general.get_camera_from_tensor (general.py:79-100): (qw,qx,qy,qz,tx,ty,tz) -> 4x4."""
    single = t.dim() == 1
    if single:
        t = t.unsqueeze(0)
    R = quad2rotation(t[:, :4])
    RT = torch.cat([R, t[:, 4:, None]], 2)
    bottom = torch.tensor([0, 0, 0, 1.0]).reshape(1, 1, 4).repeat(RT.shape[0], 1, 1)
    RT = torch.cat([RT, bottom], 1)
    return RT[0] if single else RT


# --------------------------------------------------------------------------------- sampler
def cube_far(rays_o, rays_d, bound, far_cap):
    """UniformSampler.near_far_from_cube (ray_sampler.py:23-35), `far` only."""
    tmin = (-bound - rays_o) / (rays_d + 1e-15)
    tmax = (bound - rays_o) / (rays_d + 1e-15)
    near = torch.where(tmin < tmax, tmin, tmax).max(dim=-1, keepdim=True)[0]
    far = torch.where(tmin > tmax, tmin, tmax).min(dim=-1, keepdim=True)[0]
    far = torch.where(far < near, torch.full_like(far, 1e9), far)
    return torch.clamp(far, max=far_cap)


def uniform_z(cfg: RenderConfig, rays_d, rays_o, training, t_rand=None):
    """UniformSampler.get_z_vals with take_sphere_intersection=True (ray_sampler.py:37-61)."""
    far_cap = 2.0 * cfg.scene_bounding_sphere * 1.75
    far = cube_far(rays_o, rays_d, cfg.scene_bounding_sphere, far_cap)
    near = cfg.near * torch.ones(rays_d.shape[0], 1)
    t = torch.linspace(0.0, 1.0, steps=cfg.n_samples_eval)
    z = near * (1.0 - t) + far * t
    if training:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    return z, near, far


def importance_z(params, cfg: RenderConfig, rays_d, rays_o, voxels, training, draws, stage="fine", aux=None):
    """ImportantSampler.get_z_vals (ray_sampler.py:90-166).

    draws: 't_rand' [R,E] (training), 'extra_idx' [n_extra] long (training), 'eik_idx' [R] long.
    NB the sampler always evaluates the full (coarse+fine) SDF (ray_sampler.py:101-102 passes no stage).
    """
    rays_d, rays_o = rays_d.detach(), rays_o.detach()
    z, near, far = uniform_z(cfg, rays_d, rays_o, training, draws.get("t_rand"))
    pts = (rays_o.unsqueeze(1) + z.unsqueeze(2) * rays_d.unsqueeze(1)).reshape(-1, 3)
    with torch.no_grad():
        sdf = sdf_vals(params, cfg, pts)
        w = volume_weights(z, sdf, pts, voxels, cfg.voxel_res)
    pdf = w[..., :-1] + 1e-5
    pdf = pdf / torch.sum(pdf, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    N = cfg.n_samples
    u = torch.linspace(0.0, 1.0, steps=N).unsqueeze(0).repeat(cdf.shape[0], 1).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_b, bin_a = torch.gather(z, 1, below), torch.gather(z, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    tt = (u - cdf_b) / denom
    z_imp = bin_b + tt * (bin_a - bin_b)
    if cfg.n_samples_extra > 0:
        if training:
            idx = draws["extra_idx"]
        else:
            idx = torch.linspace(0, z.shape[1] - 1, cfg.n_samples_extra).long()
        extra = torch.cat([near, far, z[:, idx]], -1)
    else:
        extra = torch.cat([near, far], -1)
    z_all, _ = torch.sort(torch.cat([z_imp, extra], -1), -1)
    if "z_vals_override" in draws:   # tests only: continue from a given sample set (the inverse CDF is
        z_all = draws["z_vals_override"]  # ill-conditioned where the pdf sits on its 1e-5 floor)
    z_eik = torch.gather(z_all, 1, draws["eik_idx"].unsqueeze(-1))
    if aux is not None:
        aux.update(cdf=cdf, bins=z, near=near, far=far)
    return z_all, z_eik


# ---------------------------------------------------------------------------------- render
def render(params: Dict[str, torch.Tensor], cfg: RenderConfig, uv, pose, intrinsics, voxels, draws,
           mode="tracking", stage="fine", color_stage="highfreq", training=True):
    """SLAMNetwork.forward (network.py:78-347) without the flow / patch-warp blocks.

    Returns the reference's output dict (+ 'gradients', 'voxels').  draws additionally holds, for
    training-mode mapping: 'eik_uniform' [10*n,3] in [-bound,bound], 'eik_jitter' [22*n... ,3] in [0,1).
    """
    ray_dirs, cam_loc = camera_rays(uv, pose, intrinsics)
    eye = torch.eye(4)[None].repeat(pose.shape[0], 1, 1)
    depth_scale = camera_rays(uv, eye, intrinsics)[0][:, :, 2:]
    bs, n_pix, _ = ray_dirs.shape
    cam = cam_loc.unsqueeze(1).repeat(1, n_pix, 1).reshape(-1, 3)
    dirs = ray_dirs.reshape(-1, 3)
    aux = {}
    z, z_eik = importance_z(params, cfg, dirs, cam, voxels, training, draws, aux=aux)
    S = z.shape[1]
    pts = (cam.unsqueeze(1) + z.unsqueeze(2) * dirs.unsqueeze(1)).reshape(-1, 3)
    out_voxels = voxels
    if mode == "mapping":
        out_voxels = update_voxels(voxels, pts.detach(), cfg.voxel_res)
        voxels = out_voxels  # in-place index_add_ on shared storage: density sees the NEW counts (network.py:72-76)
    dirs_flat = dirs.unsqueeze(1).repeat(1, S, 1).reshape(-1, 3)
    sdf, feat, grads = sdf_outputs(params, cfg, pts, stage)
    rgb = colour_net(params, cfg, pts, grads, dirs_flat, feat, color_stage).reshape(-1, S, 3)
    w = volume_weights(z, sdf, pts, voxels, cfg.voxel_res)
    rgb_values = torch.sum(w.unsqueeze(-1) * rgb, 1)
    depth = torch.sum(w * z, 1, keepdims=True) / (w.sum(dim=1, keepdims=True) + 1e-8)
    depth_values = depth_scale * depth.reshape(bs, -1, 1)
    out = {
        "rgb": rgb,
        "rgb_values": rgb_values.reshape(bs, -1, 3),
        "depth_values": depth_values,
        "z_vals": z,
        "depth_vals": z * depth_scale.reshape(-1, 1),
        "sdf": sdf.reshape(z.shape),
        "weights": w,
        "entropy": (-w * torch.log(w + 1e-4)).sum(dim=-1).mean(),
        "gradients": grads,
        "voxels": out_voxels,
        "sampler_cdf": aux["cdf"],
        "sampler_bins": aux["bins"],
    }
    if training and "vis" not in mode and "mapping" in mode:
        n = bs * n_pix
        bound = cfg.scene_bounding_sphere
        eik = draws["eik_uniform"]
        with torch.no_grad():
            near_pts = (cam.unsqueeze(1) + z_eik.unsqueeze(2) * dirs.unsqueeze(1)).reshape(-1, 3)
        eik = torch.cat([eik, near_pts], 0)
        nei = eik + (draws["eik_jitter"] - 0.5) * 0.01
        eik = torch.cat([eik, nei], 0)
        gt = sdf_gradient(params, cfg, eik, stage)
        out["grad_theta"] = gt[: gt.shape[0] // 2]
        out["grad_theta_nei"] = gt[gt.shape[0] // 2:]
        assert draws["eik_uniform"].shape[0] == 10 * n and abs(bound) > 0
    normals = grads / (grads.norm(2, -1, keepdim=True) + 1e-6)
    nmap = torch.sum(w.unsqueeze(-1) * normals.reshape(-1, S, 3), 1).reshape(bs, -1, 3)
    out["normal_map"] = torch.einsum("bij,bni->bnj", pose[:, :3, :3], nmap)
    return out


def cdf_at(z, bins, cdf):
    """Piecewise-linear sampler CDF evaluated at z (tests: compare sample sets in u-space)."""
    idx = torch.clamp(torch.searchsorted(bins.contiguous(), z.contiguous(), right=True) - 1, 0, bins.shape[1] - 2)
    b0, b1 = torch.gather(bins, 1, idx), torch.gather(bins, 1, idx + 1)
    c0, c1 = torch.gather(cdf, 1, idx), torch.gather(cdf, 1, idx + 1)
    t = torch.clamp((z - b0) / torch.clamp(b1 - b0, min=1e-20), 0, 1)
    return c0 + t * (c1 - c0)


def rgb_l1(out, rgb_gt):
    """SLAMLoss.get_rgb_loss with torch.nn.L1Loss (loss.py:57-65,131) -- the tracking objective."""
    return (out["rgb_values"].reshape(-1, 3) - rgb_gt.reshape(-1, 3)).abs().mean()
