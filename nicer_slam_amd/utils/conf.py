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


def replica_model_conf(n_samples=64, n_samples_eval=640, n_samples_extra=32, **overrides):
    """The ``model { ... }`` subtree shared by all 23 shipped run configs (e.g.
    code/confs/replica/runconf_replica_1.conf:73-160), as a dict."""
    def sdf(dims, end_size, num_levels, level_dim):
        return dict(d_in=3, d_out=1, dims=dims, geometric_init=True, bias=0.6, skip_in=[], weight_norm=True,
                    multires=6, inside_outside=True, use_grid_feature=True, base_size=32, end_size=end_size,
                    logmap=19, num_levels=num_levels, level_dim=level_dim, divide_factor=1.0,
                    embedding_method="nerf")
    conf = dict(
        feature_vector_size=64, scene_bounding_sphere=1.0, use_warp_loss=True, mapping_patchsizes=[1],
        tracking_patchsizes=[1], sampling_method="important", density_method="volsdf_gridpredefined",
        implicit_network=dict(coarse=sdf([64], 32, 4, 8), fine=sdf([64, 64, 64], 128, 8, 4)),
        rendering_network=dict(mode="idr", d_in=9, d_out=3, dims=[64, 64], weight_norm=True, multires_view=4,
                               per_image_code=False, use_grid_feature=True),
        density=dict(params_init=dict(beta=0.1), beta_min=0.0001),
        gridpredefinedensity={},
        ray_sampler=dict(near=0.0, N_samples=n_samples, N_samples_eval=n_samples_eval,
                         N_samples_extra=n_samples_extra),
    )
    conf.update(overrides)
    return Conf(conf)
