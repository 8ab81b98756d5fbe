"""Minimal stand-in for the pyhocon ConfigTree accessors the model layer uses (get_int/float/bool/string/list/
config with ``default=``), so a plain nested dict can configure the model when pyhocon is not installed.
A real pyhocon tree is used as-is."""

_MISSING = object()


class Conf(dict):
    def _get(self, key, default=_MISSING):
        cur = self
        for part in key.split("."):
            if not isinstance(cur, dict) or part not in cur:
                if default is _MISSING:
                    raise KeyError(key)
                return default
            cur = cur[part]
        return cur

    def get_int(self, key, default=_MISSING):
        return int(self._get(key, default))

    def get_float(self, key, default=_MISSING):
        return float(self._get(key, default))

    def get_bool(self, key, default=_MISSING):
        return bool(self._get(key, default))

    def get_string(self, key, default=_MISSING):
        return str(self._get(key, default))

    def get_list(self, key, default=_MISSING):
        return list(self._get(key, default))

    def get_config(self, key, default=_MISSING):
        return Conf(self._get(key, default))


def as_conf(obj):
    return obj if hasattr(obj, "get_config") else Conf(obj)


# What differs between the three shipped run-config families in the ``model { ... }`` subtree (everything else is identical in
# all 23 files): the coarse SDF network's sphere radius and whether the fine SDF network starts from the geometric initialisation
# (code/confs/replica/runconf_replica_1.conf:96,118 vs code/confs/7scenes/runconf_7scenes_1.conf:98,122 = azure/runconf_azure_*.conf).
_FAMILY_SDF = {
    "replica": dict(coarse_bias=0.6, fine_geometric_init=True),
    "7scenes": dict(coarse_bias=1.0, fine_geometric_init=False),
    "azure": dict(coarse_bias=1.0, fine_geometric_init=False),
}


def model_conf(family="replica", n_samples=64, n_samples_eval=640, n_samples_extra=32, **overrides):
    """The ``model { ... }`` subtree of the shipped run configs of `family` ("replica": code/confs/replica/*.conf and runconf_demo_2.conf;
    "7scenes": code/confs/7scenes/*.conf; "azure": code/confs/azure/*.conf and runconf_demo_1.conf), as a dict.  Held key by key against
    all 23 shipped files by tests/test_model_cpu.py::test_conf_presets_equal_the_shipped_run_configs."""
    fam = _FAMILY_SDF[family]

    def sdf(dims, end_size, num_levels, level_dim, geometric_init, bias, **extra):
        return dict(d_in=3, d_out=1, dims=dims, geometric_init=geometric_init, bias=bias, skip_in=[], weight_norm=True,
                    multires=6, inside_outside=True, use_grid_feature=True, base_size=32, end_size=end_size,
                    logmap=19, num_levels=num_levels, level_dim=level_dim, divide_factor=1.0,
                    embedding_method="nerf", **extra)
    # (the 7-Scenes / Azure files spell out `concat_coarse_feature = false` and, for the fine network, `clamp = false`: the
    # constructor defaults)
    explicit = {} if family == "replica" else dict(concat_coarse_feature=False)
    conf = dict(
        feature_vector_size=64, scene_bounding_sphere=1.0, use_warp_loss=True, mapping_patchsizes=[1],
        tracking_patchsizes=[1], sampling_method="important", density_method="volsdf_gridpredefined",
        implicit_network=dict(coarse=sdf([64], 32, 4, 8, True, fam["coarse_bias"], **explicit),
                              fine=sdf([64, 64, 64], 128, 8, 4, fam["fine_geometric_init"], 0.6,
                                       **(dict(explicit, clamp=False) if explicit else {}))),
        rendering_network=dict(mode="idr", d_in=9, d_out=3, dims=[64, 64], weight_norm=True, multires_view=4,
                               per_image_code=False, use_grid_feature=True),
        density=dict(params_init=dict(beta=0.1), beta_min=0.0001),
        gridpredefinedensity={},
        ray_sampler=dict(near=0.0, N_samples=n_samples, N_samples_eval=n_samples_eval,
                         N_samples_extra=n_samples_extra),
    )
    conf.update(overrides)
    return Conf(conf)


def replica_model_conf(n_samples=64, n_samples_eval=640, n_samples_extra=32, **overrides):
    """The ``model { ... }`` subtree of the Replica run configs and runconf_demo_2.conf (e.g. code/confs/replica/runconf_replica_1.conf:73-160)."""
    return model_conf("replica", n_samples, n_samples_eval, n_samples_extra, **overrides)


def scenes7_model_conf(n_samples=64, n_samples_eval=640, n_samples_extra=32, **overrides):
    """The ``model { ... }`` subtree of the 7-Scenes run configs (code/confs/7scenes/runconf_7scenes_1.conf:77-162; the Azure
    files hold the same subtree): coarse sphere radius 1.0, fine SDF network at nn.Linear's default initialisation."""
    return model_conf("7scenes", n_samples, n_samples_eval, n_samples_extra, **overrides)


# The rest of a run config that the hot path's callers read: image size, camera, loss weights, loop counts (the values every file
# of a family shares; per-scene keys -- scan_id, n_images, expname -- are not listed; the two demo files run 30 / 50 iterations per
# frame and 4096 mapping pixels instead of 100 / 8192).
_LOOP = dict(mapping_window_size=15, BA=True, BA_ratio=0.7, BA_cam_lr=0.001, keyframe_every=10, mapping_every_frame=5,
             mapping_iters=100, tracking_lr=0.005, tracking_iters=100, learning_rate=0.002, lr_factor_for_coarse_grid=20.0,
             lr_factor_for_fine_grid=20.0, lr_factor_for_color_grid=5.0, tracking_num_pixels=1024, mapping_num_pixels=8192)
_LOSS = dict(assign_scale_shift_init=True, warp_loss_weight=0.5, warp_loss_type="l1", rgb_loss="torch.nn.L1Loss",
             eikonal_weight=0.1, smooth_weight=0.005, depth_weight=0.1, normal_l1_weight=0.05, normal_cos_weight=0.05,
             flow_weight=0.001)
RUN_CONFS = {
    # code/confs/replica/runconf_replica_1.conf:1-72; camera of preprocess/replica_2_volsdf.py:69-73
    # (no shipped file sets SLAM.tracking.const_speed_assumption = true; volsdf_train.py:32 defaults it to False)
    "replica": dict(_LOOP, img_res=(680, 1200), intrinsics=(600.0, 600.0, 599.5, 339.5), const_speed_assumption=False,
                    loss=dict(_LOSS),
                    data_dir="../Datasets/processed/Replica", gt_traj="gt_replica_room0.txt"),
    # code/confs/7scenes/runconf_7scenes_1.conf:1-76 (smooth_weight 0.05; const_speed_assumption = false; camera of
    # preprocess/get_mesh_7scenes.py:37: fx, fy, cx, cy = 585, 585, 320, 240)
    "7scenes": dict(_LOOP, img_res=(480, 640), intrinsics=(585.0, 585.0, 320.0, 240.0), const_speed_assumption=False,
                    loss=dict(_LOSS, smooth_weight=0.05), data_dir="../Datasets/processed/7Scenes",
                    gt_traj="gt_7scenes_office.txt"),
    # code/confs/azure/runconf_azure_2.conf:1-79 (assign_scale 15).  The Azure sequences' intrinsics come out of COLMAP per sequence
    # (preprocess/azure_2_volsdf.py:62-64): the camera below is a NOMINAL Azure Kinect colour camera at 720p for synthetic stand-ins only
    "azure": dict(_LOOP, img_res=(720, 1280), intrinsics=(607.0, 607.0, 639.5, 359.5), const_speed_assumption=False,
                  loss=dict(_LOSS, assign_scale=15.0),
                  data_dir="../Datasets/processed/Azure", gt_traj="gt_azure_2.txt"),
}


def run_conf(family="replica"):
    """image size / camera / loss weights / loop counts of the shipped run configs of `family` (a copy)."""
    import copy
    return copy.deepcopy(RUN_CONFS[family])
