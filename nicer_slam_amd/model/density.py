"""SDF -> density (reference code/model/density.py:16-67)."""
import torch
import torch.nn as nn

# constants of GridPredefineDensity.func (density.py:56-59)
BETA_A, BETA_B, BETA_C, BETA_D = 0.01207724805, 0.0116544676, 0.0023639156, 5.37538


def laplace_density(sdf, beta):
    alpha = 1 / beta
    return alpha * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))


class LaplaceDensity(nn.Module):
    """alpha * Laplace(0, beta).cdf(-sdf) with a learnable beta (density.py:16-29)."""

    def __init__(self, params_init={}, beta_min=0.0001):
        super().__init__()
        for k, v in params_init.items():
            setattr(self, k, nn.Parameter(torch.tensor(v)))
        self.register_buffer("beta_min", torch.tensor(beta_min), persistent=False)

    def get_beta(self, x=None):
        return self.beta.abs() + self.beta_min

    def density_func(self, sdf, beta=None, x=None):
        return laplace_density(sdf, self.get_beta() if beta is None else beta)

    def forward(self, sdf, beta=None, x=None):
        return self.density_func(sdf, beta=beta, x=x)


class GridPredefineDensity(nn.Module):
    """beta looked up from the 64^3 visit counter (density.py:33-67).  ``voxels``/``voxel_res`` are attached
    by SLAMNetwork exactly like the reference (network.py:57-60)."""

    def __init__(self):
        super().__init__()
        self.voxels = None
        self.voxel_res = 64

    def func(self, x):
        outside = (x.abs() > 0.99).any(dim=1)
        idx = ((x + 1) / 2 * self.voxel_res).long().clamp_(0, self.voxel_res - 1)  # clamp only matters for `outside`
        count = self.voxels[idx[:, 0], idx[:, 1], idx[:, 2]]
        count = torch.where(outside, torch.zeros_like(count), count)
        return BETA_A * torch.exp(-BETA_B * 0.0001 * count * BETA_D) + BETA_C

    def get_beta(self, x):
        return self.func(x).unsqueeze(-1)

    def density_func(self, sdf, beta=None, x=None):
        return laplace_density(sdf, self.get_beta(x) if beta is None else beta)

    def forward(self, sdf, x=None, beta=None):
        return self.density_func(sdf, x=x, beta=beta)
