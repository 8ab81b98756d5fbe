"""SLAMNetwork -- the model plug-in (reference code/model/network.py:14-370).

Select it with ``train.model_class = nicer_slam_amd.model.network.SLAMNetwork``: same constructor
``(conf, dataset, n_images)``, same ``forward(input, indices, ground_truth, keyframe_list, frame_idx, mode, stage,
color_stage, iter) -> dict`` keys, same sub-module / parameter names, same ``voxels`` / ``density.voxels``
attributes, so code/training/volsdf_train.py drives it unchanged.

Two engines produce the same numbers:
  * ``fused``    -- hand-written gfx950 kernels for the whole per-frame core (rays -> sampler -> encoders/MLPs ->
                    composite -> gradients), used whenever the configuration is in the compiled set;
  * ``composed`` -- torch ops around the HIP hash-grid operator, covering every option of the module layer.
Both require the HIP extension; there is no CPU path.
"""
import torch
import torch.nn as nn

from ..utils import rend_util
from ..utils.conf import as_conf
from ..utils.general import index_to_1d
from .base_networks import ImplicitNetworkGrid_COMBINE, RenderingNetwork
from .density import GridPredefineDensity, LaplaceDensity
from .ray_sampler import ImportantSampler, transmittance_weights


class SLAMNetwork(nn.Module):
    def __init__(self, conf, dataset=None, n_images=2000, colour_grid=None):
        super().__init__()
        conf = as_conf(conf)
        self.dataset = dataset
        self.H, self.W = dataset.img_res if dataset is not None else (0, 0)
        self.white_bkgd = conf.get_bool("white_bkgd", default=False)
        self.feature_vector_size = conf.get_int("feature_vector_size")
        self.use_warp_loss = conf.get_bool("use_warp_loss", default=False)
        self.embedding_method = conf.get_string("embedding_method", default="nerf")
        self.mapping_patchsizes = conf.get_list("mapping_patchsizes", default=[1, 5, 11])
        self.tracking_patchsizes = conf.get_list("tracking_patchsizes", default=[1, 5, 11])
        self.scene_bounding_sphere = conf.get_float("scene_bounding_sphere", default=1.0)
        self.register_buffer("bg_color", torch.tensor(conf.get_list("bg_color", default=[1.0, 1.0, 1.0])).float(),
                             persistent=False)
        self.implicit_network = ImplicitNetworkGrid_COMBINE(
            conf.get_config("implicit_network"), self.feature_vector_size,
            0.0 if self.white_bkgd else self.scene_bounding_sphere)
        self.rendering_network = RenderingNetwork(
            self.feature_vector_size, n_images=n_images, embedding_method=self.embedding_method,
            colour_grid=colour_grid, **conf.get_config("rendering_network"))
        self.density_method = conf.get_string("density_method", default="volsdf_gridpredefined")
        if self.density_method == "volsdf_laplace":
            self.density = LaplaceDensity(**conf.get_config("density"))
        elif self.density_method == "volsdf_gridpredefined":
            self.density = GridPredefineDensity(**conf.get_config("gridpredefinedensity"))
        else:
            raise NotImplementedError(self.density_method)
        self.sampling_method = conf.get_string("sampling_method", default="important")
        if self.sampling_method != "important":
            raise NotImplementedError(self.sampling_method)
        self.ray_sampler = ImportantSampler(self.scene_bounding_sphere, **conf.get_config("ray_sampler"))
        # visit counter (network.py:54-60): a plain tensor attribute, shared with the density module
        self.voxel_res = conf.get_int("voxel_res", default=64)
        self.voxels = torch.zeros((self.voxel_res,) * 3)
        self.voxels_shape = self.voxels.shape
        self._share_voxels()
        self.draws = None          # optional dict of pre-drawn randoms (parity tests); else device generator
        self.engine = "auto"       # "auto" | "fused" | "composed"
        self.last_engine = None    # which engine the most recent forward used
        self.mlp_precision = "fp32"   # fused engine only: "fp32" | "bf16" | "bf16_colour" (fused/sampler.py::precision_of)
        self.voxel_sync = None     # multi-GPU mapping: callable(voxels, before) summing the visit deltas over ranks
        # Gradients the reference's loop computes and never reads (both default to "skip"; set True for the reference's
        # literal autograd behaviour -- still on the fused engine):
        #  * tracking: volsdf_train.py:406-427 back-propagates into EVERY model parameter although only the camera is stepped;
        #    those .grad are zeroed by optimizer.zero_grad() (:547) before the next use.
        #  * fine SDF MLP: pretrained, never in the optimizer (:140-173), but left requires_grad=True.
        self.tracking_param_grads = False
        self.fine_mlp_grads = False
        self.warp_engine = "auto"   # "auto": flow / patch-warp blocks as HIP kernels on the fused engine; "torch": model/warp.py

    # ------------------------------------------------------------------ plumbing
    def _share_voxels(self):
        if "gridpredefined" in self.density_method:
            self.density.voxels = self.voxels
            self.density.voxel_res = self.voxel_res

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.voxels = fn(self.voxels)
        self._share_voxels()
        return out

    def __setattr__(self, name, value):
        super().__setattr__(name, value)
        if name == "voxels" and "density" in self._modules and isinstance(value, torch.Tensor):
            self._share_voxels()

    def draw(self, kind, spec):
        """Random draws of the render core; see model/ray_sampler.py.  ``self.draws[kind]`` wins when present."""
        dev = self.voxels.device
        if self.draws is not None and kind in self.draws:
            return self.draws[kind].to(dev)
        # default device generator: graph-capture safe (philox offsets are advanced per replay)
        if kind in ("t_rand", "eik_jitter"):
            return torch.rand(tuple(spec), device=dev)
        if kind == "extra_idx":      # k distinct indices of n, uniformly: first k of a random permutation
            n, k = spec
            return torch.rand(n, device=dev).argsort()[:k]
        if kind == "eik_idx":
            high, n = spec
            return torch.randint(high, (n,), device=dev)
        if kind == "eik_uniform":
            n, b = spec
            return (torch.rand((n, 3), device=dev) * 2 - 1) * b
        raise KeyError(kind)

    # ------------------------------------------------------------------ visit counter
    def update_voxels(self, x):
        """Count sample visits per 64^3 voxel, skipping |x_d| > 0.99 (network.py:62-76).  In place, so the
        density module (which shares the storage) sees the new counts within the same forward."""
        keep = ~(x.abs() > 0.99).any(dim=1)
        idx = ((x[keep] + 1) / 2 * self.voxel_res).long()
        flat = index_to_1d(idx, self.voxel_res)
        self.voxels.view(-1).index_add_(0, flat, torch.ones_like(flat, dtype=self.voxels.dtype))

    def freeze_fine_mlp(self):
        """The fine SDF MLP is pretrained and never handed to the optimizer (volsdf_train.py:143-173).  Optional: the
        fused engine already skips its gradients unless ``fine_mlp_grads`` is set; this states it on the parameters."""
        for p in self.implicit_network.fine.mlp_parameters():
            p.requires_grad_(False)
        return self

    def _fused_composite_ok(self, mode, ground_truth):
        """Which engine renders this call.  Returns None (composed), "data" (fused kernels, pose gradient only) or
        "params" (fused kernels with parameter gradients: mapping, and tracking when ``tracking_param_grads``)."""
        if self.engine == "composed" or not self.voxels.is_cuda:
            return None
        from ..fused import render as fused_render, mapping as fused_mapping
        needs_params = (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
                        and (mode != "tracking" or self.tracking_param_grads))
        kind = None
        if fused_render.supported(self):
            if not needs_params:
                kind = "data"
            elif fused_mapping.params_supported(self):
                kind = "params"
        if kind is None and self.engine == "fused":
            raise RuntimeError("engine='fused' requested but this call is outside the fused engine's coverage "
                               "(unsupported configuration, or a parameter other than the grid tables and the three MLPs "
                               "requires grad)")
        return kind

    # ------------------------------------------------------------------ forward
    def forward(self, input, indices, ground_truth, keyframe_list=None, frame_idx=-1, mode="vis", stage="fine",
                color_stage="highfreq", iter=0):
        if mode == "tracking":
            self.patchsizes = self.tracking_patchsizes
        elif mode == "mapping":
            self.patchsizes = self.mapping_patchsizes
        intrinsics, uv, pose = input["intrinsics"], input["uv"], input["pose"]
        self.last_engine = "composed"
        self.__dict__.pop("_flat_inputs_fwd", None)
        bs, num_pixels, _ = uv.shape
        fused_kind = self._fused_composite_ok(mode, ground_truth)
        fused = fused_kind is not None
        graphed = False
        if fused:
            # an unmodified tracking loop (volsdf_train.py:406-443): rays .. composite and their backward as two cached hipGraphs
            # behind one autograd.Function (fused/track_graph.py) -- same dict, same gradient on input["pose"]
            from ..fused import track_graph
            graphed = track_graph.usable(self, mode, fused_kind, input, ground_truth)
        if graphed:
            self.last_engine = "fused"
            return track_graph.render(self, input, stage, color_stage, ground_truth)
        if fused and pose.shape[1] == 4 and uv.dtype == torch.float32:
            from ..fused import render as fused_render
            cam_flat, dirs, ds_flat = fused_render.rays(pose, uv, intrinsics.to(uv.device))
            depth_scale = ds_flat.reshape(bs, num_pixels, 1)
        else:
            ray_dirs, cam_loc = rend_util.get_camera_params(uv, pose, intrinsics)
            eye = torch.eye(4, device=pose.device, dtype=pose.dtype)[None].repeat(pose.shape[0], 1, 1)
            depth_scale = rend_util.get_camera_params(uv, eye, intrinsics)[0][:, :, 2:]   # network.py:99-102
            cam_flat = cam_loc.unsqueeze(1).repeat(1, num_pixels, 1).reshape(-1, 3)
            dirs = ray_dirs.reshape(-1, 3)

        z_vals, z_samples_eik = self.ray_sampler.get_z_vals(dirs, cam_flat, self, frame_idx, keyframe_list, mode)
        if self.draws is not None and "z_vals_override" in self.draws:   # parity tests only
            z_vals = self.draws["z_vals_override"].to(z_vals.device)
            if "eik_idx" in self.draws and z_samples_eik is not None:    # the near-surface sample is one of z_vals
                z_samples_eik = torch.gather(z_vals, 1, self.draws["eik_idx"].to(z_vals.device).unsqueeze(-1))
        N = z_vals.shape[1]
        if not fused:
            points_flat = (cam_flat.unsqueeze(1) + z_vals.unsqueeze(2) * dirs.unsqueeze(1)).reshape(-1, 3)
        if mode == "mapping":
            before = self.voxels.clone() if self.voxel_sync is not None else None
            if fused:
                from ..fused import mapping as fused_mapping
                fused_mapping.update_voxels(self, cam_flat.detach(), dirs.detach(), z_vals.detach())
            else:
                self.update_voxels(points_flat.detach())
            if before is not None:
                self.voxel_sync(self.voxels, before)

        if fused:
            from ..fused import render as fused_render, mapping as fused_mapping
            self.last_engine = "fused"
            engine = fused_mapping if fused_kind == "params" else fused_render
            if fused_kind == "params":
                self.__dict__["_flat_inputs_fwd"] = None      # composite + eikonal pass of this call share their autograd inputs
            rgb_values, depth, nmap_w, weights, ent_ray, sdf, rgb, gradients = engine.composite(
                self, cam_flat, dirs, z_vals, stage, color_stage)
        else:
            dirs_flat = dirs.unsqueeze(1).repeat(1, N, 1).reshape(-1, 3)
            sdf, feats, gradients = self.implicit_network.get_outputs(points_flat, stage=stage)
            rgb = self.rendering_network(points_flat, gradients, dirs_flat, feats, indices,
                                         color_stage=color_stage).reshape(-1, N, 3)
            weights = self.volume_rendering(z_vals, sdf, points_flat)
            rgb_values = torch.sum(weights.unsqueeze(-1) * rgb, 1)
            depth = torch.sum(weights * z_vals, 1, keepdims=True) / (weights.sum(dim=1, keepdims=True) + 1e-8)
            nmap_w, ent_ray = None, None
        try:
            return self._assemble(mode, bs, num_pixels, uv, pose, intrinsics, ground_truth, stage, fused, fused_kind, depth_scale,
                                  cam_flat, dirs, z_vals, z_samples_eik, rgb_values, depth, nmap_w, weights, ent_ray, sdf, rgb,
                                  gradients)
        finally:
            self.__dict__.pop("_flat_inputs_fwd", None)

    def _assemble(self, mode, bs, num_pixels, uv, pose, intrinsics, ground_truth, stage, fused, fused_kind, depth_scale, cam_flat,
                  dirs, z_vals, z_samples_eik, rgb_values, depth, nmap_w, weights, ent_ray, sdf, rgb, gradients):
        """The forward's output dict from the renderer's results (network.py:153-165, 281-345)."""
        N = z_vals.shape[1]
        output = {}
        # keyframe re-projection blocks: HIP kernels on the fused engine (fused/warp.py), torch ops otherwise (model/warp.py)
        warp_kernels = False
        if fused and self.warp_engine != "torch":
            from ..fused import warp as fused_warp
            warp_kernels = fused_warp.available(depth, uv, pose)
        if "edges" in ground_truth:   # optical-flow reprojection (network.py:153-165)
            if warp_kernels:
                output["flow"] = fused_warp.flow(self, uv, pose, intrinsics, depth, ground_truth["edges"])
            else:
                from .warp import flow_reproject
                output["flow"] = flow_reproject(uv, pose, intrinsics, depth, ground_truth["edges"])
        if self.use_warp_loss and ("vis" not in mode) and ("tracking" not in mode):
            if warp_kernels:
                output["warp_output"] = fused_warp.patch_warp(self, uv, pose, intrinsics, depth, ground_truth, bs)
            else:
                from .warp import patch_warp
                output["warp_output"] = patch_warp(self, uv, pose, intrinsics, depth.unsqueeze(2), ground_truth, bs)

        depth_values = depth_scale * depth.reshape(bs, -1, 1)
        if self.white_bkgd:
            rgb_values = rgb_values + (1.0 - weights.sum(-1)[..., None]) * self.bg_color.unsqueeze(0)
        output.update({
            "rgb": rgb,
            "rgb_values": rgb_values.reshape(bs, -1, 3),
            "depth_values": depth_values,
            "z_vals": z_vals,
            "depth_vals": z_vals * depth_scale.reshape(-1, 1),
            "sdf": sdf.reshape(z_vals.shape),
            "weights": weights,
            "entropy": ent_ray.mean() if fused else (-weights * torch.log(weights + 1e-4)).sum(dim=-1).mean(),
            "scene_bounding_sphere": self.scene_bounding_sphere,
        })
        if self.training and ("vis" not in mode) and ("mapping" in mode):   # eikonal samples (network.py:313-336)
            n = bs * num_pixels
            eik = self.draw("eik_uniform", (n * 10, self.scene_bounding_sphere))
            with torch.no_grad():
                near_surface = (cam_flat.unsqueeze(1) + z_samples_eik.unsqueeze(2) * dirs.unsqueeze(1)).reshape(-1, 3)
            eik = torch.cat([eik, near_surface], 0)
            eik = torch.cat([eik, eik + (self.draw("eik_jitter", eik.shape) - 0.5) * 0.01], 0)
            if fused_kind == "params":
                from ..fused import mapping as fused_mapping
                output["grad_theta"], output["grad_theta_nei"] = fused_mapping.sdf_gradient(self, eik, stage, halves=True)
            else:
                grad_theta = self.implicit_network.gradient(eik, stage=stage)
                half = grad_theta.shape[0] // 2
                output["grad_theta"], output["grad_theta_nei"] = grad_theta[:half], grad_theta[half:]
        if fused:
            normal_map = nmap_w.reshape(bs, -1, 3)
        else:
            normals = gradients / (gradients.norm(2, -1, keepdim=True) + 1e-6)
            normal_map = torch.sum(weights.unsqueeze(-1) * normals.reshape(-1, N, 3), 1).reshape(bs, -1, 3)
        # einsum("bij,bni->bnj", R, n) (network.py:345) = n @ R per camera: the same batched GEMM without einsum's host-side planning
        output["normal_map"] = torch.matmul(normal_map, pose[:, :3, :3])
        return output

    def volume_rendering(self, z_vals, sdf, points_flat, rays_o=None, rays_d=None, gradients=None, frame_idx=1,
                         mode=None):
        """SDF -> density -> alpha/transmittance weights (network.py:349-370)."""
        density = self.density(sdf, x=points_flat).reshape(-1, z_vals.shape[1])
        return transmittance_weights(z_vals, density)
