"""SDF and colour networks (reference code/model/base_networks.py:7-405) -- module layer.

Parameter/attribute names match the reference so its checkpoints load and its optimizer-group code
(code/training/volsdf_train.py:150-173) works unchanged: ``encoding.embeddings``, ``lin{i}.weight_g/weight_v/bias``,
``grid_parameters()``, ``mlp_parameters()``.  ``forward`` here is the composed (torch-op + HIP hash kernels)
engine; SLAMNetwork routes supported configurations to the fused HIP engine instead.
"""
import numpy as np
import torch
import torch.nn as nn

from ..hashencoder.hashgrid import HashEncoder
from .embedder import get_embedder


class ImplicitNetworkGrid(nn.Module):
    """Hash-grid feature (+) positional encoding -> weight-normalised Softplus(beta=100) MLP -> [sdf, feature]."""

    def __init__(self, feature_vector_size, sdf_bounding_sphere, d_in, d_out, dims, geometric_init=True, bias=1.0,
                 skip_in=(), weight_norm=True, multires=0, sphere_scale=1.0, inside_outside=False, base_size=16,
                 end_size=2048, logmap=19, num_levels=16, level_dim=2, embedding_method="nerf", divide_factor=1.5,
                 use_grid_feature=True, name="", clamp=False, concat_coarse_feature=False):
        super().__init__()
        if concat_coarse_feature:
            raise NotImplementedError("concat_coarse_feature is not used by any shipped config")
        self.concat_coarse_feature = False
        self.name = name
        self.sdf_bounding_sphere = sdf_bounding_sphere
        self.sphere_scale = sphere_scale
        self.divide_factor = divide_factor
        self.grid_feature_dim = num_levels * level_dim
        self.use_grid_feature = use_grid_feature
        self.clamp = clamp
        self.multires = multires
        self.skip_in = tuple(skip_in)
        dims = [d_in] + list(dims) + [d_out + feature_vector_size]
        dims[0] += self.grid_feature_dim
        self.encoding = HashEncoder(input_dim=3, num_levels=num_levels, level_dim=level_dim, per_level_scale=2,
                                    base_resolution=base_size, log2_hashmap_size=logmap, desired_resolution=end_size)
        self.embed_fn = None
        if multires > 0:
            self.embed_fn, input_ch = get_embedder(multires, input_dims=d_in, embed_type=embedding_method)
            dims[0] += input_ch - 3
        self.num_layers = len(dims)
        self.dims = dims
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            lin = nn.Linear(dims[l], out_dim)
            if geometric_init:   # base_networks.py:127-146
                if l == self.num_layers - 2:
                    sign = -1.0 if inside_outside else 1.0
                    nn.init.normal_(lin.weight, mean=sign * np.sqrt(np.pi) / np.sqrt(dims[l]), std=0.0001)
                    nn.init.constant_(lin.bias, bias if inside_outside else -bias)
                elif multires > 0 and l == 0:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.constant_(lin.weight[:, 3:], 0.0)
                    nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
                elif multires > 0 and l in self.skip_in:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
                    nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
                else:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            if weight_norm:
                lin = nn.utils.weight_norm(lin)
            setattr(self, "lin" + str(l), lin)
        self.softplus = nn.Softplus(beta=100)

    def forward(self, input, c_feature_vectors=None):
        if self.use_grid_feature:
            feature = self.encoding(input / self.divide_factor)
        else:
            feature = input.new_zeros(input.shape[0], self.grid_feature_dim)
        first = torch.cat((self.embed_fn(input) if self.embed_fn is not None else input, feature), dim=-1)
        x = first
        for l in range(self.num_layers - 1):
            if l in self.skip_in:
                x = torch.cat([x, first], 1) / np.sqrt(2)
            x = getattr(self, "lin" + str(l))(x)
            if l < self.num_layers - 2:
                x = self.softplus(x)
        if self.clamp and self.name == "fine":
            x = torch.cat([torch.tanh(x[:, :1]) * 0.05, x[:, 1:]], dim=-1)
        return x

    def get_feature(self, x, c_feature_vectors=None, stage=None):
        return self.forward(x)[:, 1:]

    def get_sdf_vals(self, x, c_feature_vectors=None, stage=None):
        return self.forward(x)[:, :1]

    def gradient(self, x, c_feature_vectors=None, stage=None):
        x.requires_grad_(True)
        y = self.forward(x)[:, :1]
        return torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True, retain_graph=True)[0]

    def get_outputs(self, x, c_feature_vectors=None, stage=None):
        x.requires_grad_(True)
        out = self.forward(x)
        sdf, feats = out[:, :1], out[:, 1:]
        grads = torch.autograd.grad(sdf, x, torch.ones_like(sdf), create_graph=True, retain_graph=True)[0]
        return sdf, feats, grads

    def mlp_parameters(self):
        out = []
        for l in range(self.num_layers - 1):
            out += list(getattr(self, "lin" + str(l)).parameters())
        return out

    def grid_parameters(self):
        return self.encoding.parameters()


class ImplicitNetworkGrid_COMBINE(nn.Module):
    """coarse + fine residual combination (base_networks.py:7-47)."""

    def __init__(self, conf, feature_vector_size, sdf_bounding_sphere):
        super().__init__()
        self.feature_vector_size = feature_vector_size
        self.sdf_bounding_sphere = sdf_bounding_sphere
        self.coarse = ImplicitNetworkGrid(feature_vector_size, sdf_bounding_sphere, name="coarse",
                                          **conf.get_config("coarse"))
        self.fine = ImplicitNetworkGrid(feature_vector_size, sdf_bounding_sphere, name="fine",
                                        **conf.get_config("fine"))

    def get_sdf_vals(self, x, stage="fine"):
        # the reference also evaluates coarse.get_feature here (base_networks.py:31) and discards it
        s = self.coarse.get_sdf_vals(x)
        return s if stage == "coarse" else s + self.fine.get_sdf_vals(x)

    def get_outputs(self, x, stage="fine"):
        c = self.coarse.get_outputs(x)
        if stage == "coarse":
            return c
        f = self.fine.get_outputs(x)
        return c[0] + f[0], c[1] + f[1], c[2] + f[2]

    def gradient(self, x, stage="fine"):
        g = self.coarse.gradient(x)
        return g if stage == "coarse" else g + self.fine.gradient(x)


_RENDER_INPUTS = {   # mode -> which of (points, view_dirs, normals, features) are concatenated, in order
    "idr": ("p", "v", "n", "f"), "idr_detach": ("p", "v", "nd", "f"), "idr_nopts": ("v", "n", "f"),
    "idr_nopts_detach": ("v", "nd", "f"), "idr_nonormal": ("p", "v", "f"), "idr_noview": ("p", "n", "f"),
    "nerf": ("v", "f"), "no_feature": ("p", "v", "n"), "no_feature_no_noraml": ("p", "v"),
}


class RenderingNetwork(nn.Module):
    """Colour MLP: [x, PE(view), grad sdf, feature, colour-grid feature] -> ReLU MLP -> sigmoid
    (base_networks.py:241-405).  The colour grid geometry is the reference's hard-coded 16x2, 16->2048, 2^24."""

    COLOUR_GRID = dict(num_levels=16, level_dim=2, base_resolution=16, desired_resolution=2048, log2_hashmap_size=24)

    def __init__(self, feature_vector_size, mode, d_in, d_out, dims, weight_norm=True, multires_view=0,
                 per_image_code=False, model_exposure=False, n_images=2000, embedding_method="nerf",
                 use_grid_feature=False, colour_grid=None):
        super().__init__()
        if model_exposure:
            raise NotImplementedError("model_exposure is not used by any shipped config")
        if mode not in _RENDER_INPUTS and mode != "no_color":
            raise ValueError(f"unknown rendering mode {mode}")
        self.use_grid_feature = use_grid_feature
        self.divide_factor = 1.0
        self.grid_feature_dim = 0
        if use_grid_feature:
            g = dict(self.COLOUR_GRID, **(colour_grid or {}))
            self.grid_feature_dim = g["num_levels"] * g["level_dim"]
            self.encoding = HashEncoder(input_dim=3, per_level_scale=2, **g)
        self.n_images = n_images
        self.mode = mode
        if mode in ("no_feature", "no_feature_no_noraml"):
            feature_vector_size = 0
        dims = [d_in + feature_vector_size + self.grid_feature_dim] + list(dims) + [d_out]
        self.embedview_fn = None
        self.multires_view = multires_view
        if multires_view > 0:
            self.embedview_fn, input_ch = get_embedder(multires_view, embed_type=embedding_method)
            dims[0] += input_ch - 3
        self.per_image_code = per_image_code
        if per_image_code:
            self.embeddings = nn.Parameter(torch.empty(n_images, 32).uniform_(-1e-4, 1e-4))
            dims[0] += 32
        self.model_exposure = False
        self.num_layers = len(dims)
        self.dims = dims
        for l in range(self.num_layers - 1):
            lin = nn.Linear(dims[l], dims[l + 1])
            if weight_norm:
                lin = nn.utils.weight_norm(lin)
            setattr(self, "lin" + str(l), lin)
        self.relu = nn.ReLU()
        self.sigmoid = nn.Sigmoid()

    def forward(self, points, normals, view_dirs, feature_vectors, indices, color_stage="base"):
        if self.mode == "no_color":
            return self.sigmoid(feature_vectors[:, :3])
        if self.embedview_fn is not None:
            view_dirs = self.embedview_fn(view_dirs)
        pick = {"p": points, "v": view_dirs, "n": normals, "nd": normals.detach(), "f": feature_vectors}
        parts = [pick[k] for k in _RENDER_INPUTS[self.mode]]
        if self.use_grid_feature and self.mode == "idr":
            gf = self.encoding(points / self.divide_factor)
            parts.append(gf.detach() if color_stage == "base" else gf)   # base_networks.py:335-339
        x = torch.cat(parts, dim=-1)
        if self.per_image_code:
            x = torch.cat([x, self.embeddings[indices].repeat(x.shape[0] // indices.shape[0], 1)], dim=-1)
        for l in range(self.num_layers - 1):
            x = getattr(self, "lin" + str(l))(x)
            if l < self.num_layers - 2:
                x = self.relu(x)
        return self.sigmoid(x)

    def mlp_parameters(self):
        out = []
        for l in range(self.num_layers - 1):
            out += list(getattr(self, "lin" + str(l)).parameters())
        return out

    def grid_parameters(self):
        return self.encoding.parameters()
