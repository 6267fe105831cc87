"""Model configurations used by the oracle, the golden generator and the tests.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Values restate the ``model{}`` block of the reference's conf files
(reference: code/confs/mi.conf:83-133,
code/confs/mp_mJXqzFtmKg4_undist_scannetMLP.conf:81-132,
code/confs/mp_mJXqzFtmKg4_undist_scannetGrids.conf:83-128) as plain dicts, per
SURVEY.md 8(d).  ``Tree`` offers the six pyhocon getters the reference's model
constructor calls (reference: code/model/network.py:481-499) so the same dict can
be handed to the real reference classes in ``make_golden.py``.
"""
import copy


class Tree(dict):
    """dict with the pyhocon ConfigTree getters the reference model uses."""

    _MISSING = object()

    def _get(self, key, default):
        if key in self:
            return self[key]
        if default is Tree._MISSING:
            raise KeyError(key)
        return default

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
        v = self._get(key, default)
        return v if isinstance(v, Tree) else Tree(v)


def _to_tree(d):
    t = Tree()
    for k, v in d.items():
        t[k] = _to_tree(v) if isinstance(v, dict) else v
    return t


_SAMPLER = dict(near=0.0, N_samples=64, N_samples_eval=128, N_samples_extra=32,
                eps=0.1, beta_iters=10, max_total_iters=5)
_RENDER = dict(mode='idr', d_in=9, d_out=3, dims=[256, 256], weight_norm=True,
               multires_view=4, per_image_code=False)
_DENSITY = dict(params_init=dict(beta=0.1), beta_min=0.0001)


def mlp_config(width=256, depth=8, beta=0.1):
    """Config 2 of BASELINE.json: upstream-style ImplicitNetwork 8x256 (Grid_MLP=False)."""
    skip = [4] if depth > 4 else []
    return _to_tree(dict(
        feature_vector_size=width, scene_bounding_sphere=1.1, Grid_MLP=False,
        implicit_network=dict(d_in=3, d_out=1, dims=[width] * depth, geometric_init=True,
                              bias=0.9, skip_in=skip, weight_norm=True, multires=6,
                              inside_outside=True),
        rendering_network=dict(_RENDER, dims=[width, width]),
        density=dict(params_init=dict(beta=beta), beta_min=0.0001),
        ray_sampler=dict(_SAMPLER)))


def gridless_config(width=256, depth=8, beta=0.1):
    """The fork's "MLP" confs: ImplicitNetworkGrid with use_grid_feature=False (71-wide input)."""
    c = mlp_config(width, depth, beta)
    c['Grid_MLP'] = True
    c['implicit_network']['use_grid_feature'] = False
    c['implicit_network']['divide_factor'] = 1.1
    return c


def grid_config(width=256, beta=0.1, num_levels=16, level_dim=2, logmap=19,
                base_size=16, end_size=2048):
    """Config 3: multi-resolution hash grid + 2x256 MLP (scannetGrids.conf:83-128)."""
    return _to_tree(dict(
        feature_vector_size=width, scene_bounding_sphere=1.1, Grid_MLP=True,
        implicit_network=dict(d_in=3, d_out=1, dims=[width, width], geometric_init=True,
                              bias=0.9, skip_in=[4], weight_norm=True, multires=6,
                              inside_outside=True, use_grid_feature=True, divide_factor=1.1,
                              num_levels=num_levels, level_dim=level_dim, logmap=logmap,
                              base_size=base_size, end_size=end_size),
        rendering_network=dict(_RENDER, dims=[width, width]),
        density=dict(params_init=dict(beta=beta), beta_min=0.0001),
        ray_sampler=dict(_SAMPLER)))


def clone(conf):
    return _to_tree(copy.deepcopy(dict(conf)))
