"""Ray samplers (reference code/model/ray_sampler.py:16-166) -- composed (torch-op) engine.

Randomness: the reference draws on the CPU generator and uploads (ray_sampler.py:58,148,158).  Here the draws
come from ``model.draw(kind, ...)`` (SLAMNetwork.draw): pre-drawn tensors when the caller supplied them
(parity tests), else the device generator."""
import torch


class UniformSampler:
    def __init__(self, scene_bounding_sphere, near, N_samples, take_sphere_intersection=False, far=-1):
        self.near = near
        self.far = 2.0 * scene_bounding_sphere * 1.75 if far == -1 else far
        self.N_samples = N_samples
        self.scene_bounding_sphere = scene_bounding_sphere
        self.take_sphere_intersection = take_sphere_intersection

    def near_far_from_cube(self, rays_o, rays_d, bound):
        """Slab test against the cube [-bound, bound]^3 (ray_sampler.py:23-35)."""
        t0 = (-bound - rays_o) / (rays_d + 1e-15)
        t1 = (bound - rays_o) / (rays_d + 1e-15)
        near = torch.minimum(t0, t1).max(dim=-1, keepdim=True)[0]
        far = torch.maximum(t0, t1).min(dim=-1, keepdim=True)[0]
        miss = far < near
        near = torch.where(miss, torch.full_like(near, 1e9), near)
        far = torch.where(miss, torch.full_like(far, 1e9), far)
        return torch.clamp(near, min=self.near), torch.clamp(far, max=self.far)

    def get_z_vals(self, ray_dirs, cam_loc, model):
        ray_dirs, cam_loc = ray_dirs.detach(), cam_loc.detach()
        n = ray_dirs.shape[0]
        near = torch.full((n, 1), float(self.near), device=ray_dirs.device, dtype=ray_dirs.dtype)
        if self.take_sphere_intersection:
            _, far = self.near_far_from_cube(cam_loc, ray_dirs, bound=self.scene_bounding_sphere)
        else:
            far = torch.full_like(near, float(self.far))
        t = torch.linspace(0.0, 1.0, steps=self.N_samples, device=ray_dirs.device)
        z = near * (1.0 - t) + far * t
        if model.training:   # stratified jitter (ray_sampler.py:52-59)
            mids = 0.5 * (z[..., 1:] + z[..., :-1])
            upper = torch.cat([mids, z[..., -1:]], -1)
            lower = torch.cat([z[..., :1], mids], -1)
            z = lower + (upper - lower) * model.draw("t_rand", z.shape)
        return z, near, far


def transmittance_weights(z_vals, density):
    """alpha_i * T_i with the last interval 1e10 (ray_sampler.py:107-112 == network.py:354-368)."""
    dists = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], z_vals.new_full((z_vals.shape[0], 1), 1e10)], -1)
    energy = dists * density
    shifted = torch.cat([energy.new_zeros(energy.shape[0], 1), energy[:, :-1]], dim=-1)
    return (1 - torch.exp(-energy)) * torch.exp(-torch.cumsum(shifted, dim=-1))


class ImportantSampler:
    def __init__(self, scene_bounding_sphere, near, N_samples, N_samples_eval, N_samples_extra,
                 inverse_sphere_bg=False, N_samples_inverse_sphere=0):
        if inverse_sphere_bg:
            raise NotImplementedError("inverse_sphere_bg is not used by any shipped config")
        self.near, self.far = near, 2.0 * scene_bounding_sphere
        self.N_samples, self.N_samples_eval, self.N_samples_extra = N_samples, N_samples_eval, N_samples_extra
        self.scene_bounding_sphere = scene_bounding_sphere
        self.uniform_sampler = UniformSampler(scene_bounding_sphere, near, N_samples_eval,
                                              take_sphere_intersection=True)

    def get_z_vals(self, ray_dirs, cam_loc, model, frame_idx=None, keyframe_list=None, mode=None):
        if getattr(model, "engine", "composed") != "composed":
            from ..fused import sampler as fused_sampler
            if fused_sampler.supported(model) and ray_dirs.is_cuda:
                model.last_engine = "fused-sampler"
                return fused_sampler.get_z_vals(model, ray_dirs, cam_loc)
            if model.engine == "fused":
                raise RuntimeError("engine='fused' requested but this configuration is outside the compiled set")
        z, near, far = self.uniform_sampler.get_z_vals(ray_dirs, cam_loc, model)
        pts = (cam_loc.unsqueeze(1) + z.unsqueeze(2) * ray_dirs.unsqueeze(1)).reshape(-1, 3)
        with torch.no_grad():                      # only the SDF evaluation is under no_grad (ray_sampler.py:101-102);
            sdf = model.implicit_network.get_sdf_vals(pts)
        # density, weights and the inverse CDF run with grad enabled like the reference (:104-139): with
        # density_method = "volsdf_laplace" the learnable beta receives a gradient through z_vals
        w = transmittance_weights(z, model.density(sdf, x=pts).reshape(z.shape))
        z_imp = inverse_cdf_samples(z, w, self.N_samples)
        if self.N_samples_extra > 0:
            if model.training:
                idx = model.draw("extra_idx", (z.shape[1], self.N_samples_extra))
            else:
                idx = torch.linspace(0, z.shape[1] - 1, self.N_samples_extra, device=z.device).long()
            extra = torch.cat([near, far, z[:, idx]], -1)
        else:
            extra = torch.cat([near, far], -1)
        z_all, _ = torch.sort(torch.cat([z_imp, extra], -1), -1)
        eik_idx = model.draw("eik_idx", (z_all.shape[-1], z_all.shape[0]))
        return z_all, torch.gather(z_all, 1, eik_idx.unsqueeze(-1))


def inverse_cdf_samples(bins, weights, n):
    """n samples at u = linspace(0,1,n) of the piecewise-linear CDF of (weights[:-1] + 1e-5)
    (ray_sampler.py:114-139)."""
    pdf = weights[..., :-1] + 1e-5
    pdf = pdf / torch.sum(pdf, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    u = torch.linspace(0.0, 1.0, steps=n, device=bins.device).unsqueeze(0).repeat(cdf.shape[0], 1).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return b0 + (u - c0) / denom * (b1 - b0)
