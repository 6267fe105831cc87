"""Drop-in model classes for the reference's ``model.network`` (code/model/network.py):
ImplicitNetwork (12-137), ImplicitNetworkGrid (141-322), RenderingNetwork (325-470) and
MonoSDFNetwork (472-640) -- same constructors, method names, output dict and state-dict keys
(``lin{l}.weight_g / weight_v / bias``, ``encoding.embeddings / offsets``, ``density.beta``).

The modules only hold parameters and orchestrate; every per-point / per-ray computation is a
HIP kernel reached through monosdf_amd.ops (no eager-PyTorch fallback: CPU tensors raise).
"""
import math

import os

import numpy as np
import torch
import torch.nn as nn

from .. import ops, plan as planlib
from ..hashencoder.hashgrid import HashEncoder
from .density import LaplaceDensity
from .ray_sampler import ErrorBoundSampler


class WNLinear(nn.Module):
    """Parameter holder with the reference's old-style weight-norm names (weight_g, weight_v, bias).

    The initial values are drawn exactly like ``nn.utils.weight_norm(nn.Linear(...))`` after the
    reference's init function ran on the Linear, so a given torch seed yields the same state dict."""

    def __init__(self, in_dim, out_dim, init_fn=None, weight_norm=True):
        super().__init__()
        lin = nn.Linear(in_dim, out_dim)          # consumes the RNG like the reference
        if init_fn is not None:
            init_fn(lin)
        self.in_features, self.out_features = in_dim, out_dim
        self.has_weight_norm = weight_norm
        self.bias = nn.Parameter(lin.bias.data.clone())
        if weight_norm:
            w = lin.weight.data
            self.weight_g = nn.Parameter(torch.norm_except_dim(w, 2, 0).clone())
            self.weight_v = nn.Parameter(w.clone())
        else:
            self.weight = nn.Parameter(lin.weight.data.clone())

    def effective_weight(self):
        if self.has_weight_norm:
            return torch._weight_norm(self.weight_v, self.weight_g, 0)
        return self.weight


class _FusedNet(nn.Module):
    """Shared plumbing: lazily created device state + per-call packing of the effective weights."""

    def _layers(self):
        return [getattr(self, 'lin%d' % l) for l in range(self.num_layers - 1)]

    # matrix core of the fused kernels: 'fp32' (default), 'bf16x3' or 'bf16x6' (ops.PRECISIONS).  Not a reference option: set it with
    # set_precision() / MonoSDFNetwork.set_precision() or the MONOSDF_PRECISION environment variable.
    precision = None
    supports_bf16x3 = False

    def set_precision(self, precision):
        if precision not in ops.PRECISIONS:
            raise ValueError('precision must be one of %s' % (ops.PRECISIONS,))
        if precision != 'fp32' and not self.supports_bf16x3:
            precision = 'fp32'
        object.__setattr__(self, 'precision', precision)
        object.__setattr__(self, '_fused_state', None)

    def _fused(self, device):
        f = getattr(self, '_fused_state', None)
        if f is None or f.device != device:
            prec = self.precision or os.environ.get('MONOSDF_PRECISION', 'fp32')
            if not self.supports_bf16x3:
                prec = 'fp32'
            f = ops.FusedMlp(self._build_plan(), device, prec)
            object.__setattr__(self, '_fused_state', f)
        return f

    def packed(self, device):
        """(fused, flat_w, flat_b, wpack, bpack); reused inside MonoSDFNetwork.forward via share()."""
        shared = getattr(self, '_shared', None)
        if shared is not None:
            return shared
        fused = self._fused(device)
        wn = getattr(self, '_wn_state', None)
        if wn is None:
            wn = ops.WeightNormState()
            object.__setattr__(self, '_wn_state', wn)
        # g * v / ||v|| of every layer in one launch (PyTorch autograd still owns weight_g / weight_v / bias)
        flat_w, flat_b = ops.fused_weight_norm(wn, self._layers())
        wpack, bpack = fused.pack(flat_w, flat_b)
        return fused, flat_w, flat_b, wpack, bpack

    def share(self, device):
        object.__setattr__(self, '_shared', None)
        object.__setattr__(self, '_shared', self.packed(device))

    def unshare(self):
        object.__setattr__(self, '_shared', None)


def _geometric_init(l, num_layers, dims, skip_in, multires, bias, inside_outside, out_dim):
    """The reference's geometric initialisation (network.py:51-70) as a function of the layer index."""
    def fn(lin):
        if l == num_layers - 2:
            if not inside_outside:
                torch.nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(dims[l]), std=0.0001)
                torch.nn.init.constant_(lin.bias, -bias)
            else:
                torch.nn.init.normal_(lin.weight, mean=-np.sqrt(np.pi) / np.sqrt(dims[l]), std=0.0001)
                torch.nn.init.constant_(lin.bias, bias)
        elif multires > 0 and l == 0:
            torch.nn.init.constant_(lin.bias, 0.0)
            torch.nn.init.constant_(lin.weight[:, 3:], 0.0)
            torch.nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
        elif multires > 0 and l in skip_in:
            torch.nn.init.constant_(lin.bias, 0.0)
            torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            torch.nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
        else:
            torch.nn.init.constant_(lin.bias, 0.0)
            torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
    return fn


class _SdfBase(_FusedNet):
    """Common part of ImplicitNetwork / ImplicitNetworkGrid."""

    supports_bf16x3 = True

    def _make_layers(self, dims, geometric_init, bias, skip_in, weight_norm, multires, inside_outside):
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        self.dims = list(dims)
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            init = _geometric_init(l, self.num_layers, dims, self.skip_in, multires, bias, inside_outside,
                                   out_dim) if geometric_init else None
            setattr(self, 'lin%d' % l, WNLinear(dims[l], out_dim, init, weight_norm))

    def _build_plan(self):
        shapes = [(l.out_features, l.in_features) for l in self._layers()]
        return planlib.build_sdf_plan(shapes, self.skip_in, self.multires, self.aux_cols, self.aux_active,
                                      self.feature_vector_size)

    # -- one fused evaluation -------------------------------------------------------------
    def evaluate(self, x, n_clamp, n_feat, save=None, split=None):
        """sdf [P,1], feat [n_feat,F], d sdf/dx [P,3] for the points x; the first n_clamp points get the
        bounding-sphere clamp, the first n_feat points get feature vectors.
        split = s: returns (sdf [:s], feat, d sdf/dx [:s], d sdf/dx [s:]) as separate tensors."""
        if save is None:
            save = torch.is_grad_enabled()
        x = x.detach()
        fused, flat_w, flat_b, wpack, bpack = self.packed(x.device)
        ns = x.shape[0] if split is None else int(split)
        if self.aux_active and ops.HASH_SCATTER == 'binned' and self.encoding.level_dim > 1:
            # encoding + MLP + grid part of d sdf/dx as one autograd node: one embedding scatter per backward pass
            enc = self.encoding
            fused.grid_offsets = enc.offsets
            sdf, _, feat, nrm, nrm_b = ops.GridSdfFunction.apply(
                x, enc.embeddings, flat_w, flat_b, wpack, bpack, fused,
                (enc.num_levels, enc.level_dim, enc.log2_scale, int(enc.base_resolution)), int(n_clamp), int(n_feat),
                self.sphere_scale, bool(save), ns, float(self.divide_factor))
            feat = self._trim_features(feat)
            return (sdf, feat, nrm) if split is None else (sdf, feat, nrm, nrm_b)
        aux, handle = None, None
        if self.aux_active:
            aux, handle = self.encoding.encode_with_jacobian((x / self.divide_factor + 1.0) / 2.0)
            aux = self._pad_aux(aux)
        radius = self.sdf_bounding_sphere if self.clamps else 0.0
        sdf, _, feat, nrm, nrm_b, r_aux = ops.SdfMlpFunction.apply(
            x, aux, flat_w, flat_b, wpack, bpack, fused, int(n_clamp), int(n_feat), radius, self.sphere_scale,
            bool(save), ns)
        if self.aux_active:
            # chain rule through x01 = (x / divide_factor + 1) / 2
            r_aux = r_aux[:, :self.aux_cols]
            through_grid = self.encoding.input_gradient(handle, r_aux) * (0.5 / self.divide_factor)
            nrm = nrm + through_grid[:ns]
            nrm_b = nrm_b + through_grid[ns:]
        feat = self._trim_features(feat)
        if split is None:
            return sdf, feat, nrm
        return sdf, feat, nrm, nrm_b

    def _trim_features(self, feat):
        """The kernels write whole 16-slot tiles; the caller gets feature_vector_size columns (a view)."""
        F = self.feature_vector_size
        return feat if feat.shape[1] == F else feat[:, :F]

    def _pad_aux(self, aux):
        """The kernels read whole 16-slot tiles of grid features."""
        pad = (-aux.shape[1]) % 16
        if pad:
            aux = torch.nn.functional.pad(aux, (0, pad))
        return aux.contiguous()

    def gradient_sdf(self, x):
        return self.evaluate(x, 0, 0)[2]

    def get_outputs(self, x):
        P = x.shape[0]
        return self.evaluate(x, P, P)

    def raw_sdf(self, x):
        """Column 0 of forward(x) -- the unclamped SDF -- without forming the 256 features or the gradient: what the
        meshing grid needs (`lambda x: implicit_network(x)[:, 0]` in the reference, monosdf_train.py:334, eval.py:76) at a
        quarter of the arithmetic and none of the 1 KB per point of feature traffic.  No gradient."""
        with torch.no_grad():
            return self._sdf_only(x, clamp=False)[:, 0]

    def _sdf_only(self, x, run_flag=None, clamp=True):
        fused, _, _, wpack, bpack = self.packed(x.device)
        aux, aux_lm = None, None
        if self.aux_active:
            enc = self.encoding
            if x.shape[-1] == 3 and x.dim() == 2:
                # x01 inside the encoder kernel; the SDF kernel reads the encoder's level-major output
                aux, aux_lm = ops.hash_node_features(
                    x, self.divide_factor, enc.embeddings, enc.offsets,
                    (enc.num_levels, enc.level_dim, enc.log2_scale, int(enc.base_resolution)), 16 * fused.plan.aux_tiles,
                    level_major=(fused.precision == 'fp32'))
            else:
                with torch.no_grad():
                    aux = self._pad_aux(enc((x / self.divide_factor).detach(), calc_grad_inputs=False))
        radius = self.sdf_bounding_sphere if (self.clamps and clamp) else 0.0
        return ops.sdf_forward_nograd(fused, wpack, bpack, x.detach(), aux, radius, self.sphere_scale, run_flag, aux_lm)


class ImplicitNetwork(_SdfBase):
    def __init__(self, feature_vector_size, sdf_bounding_sphere, d_in, d_out, dims, geometric_init=True,
                 bias=1.0, skip_in=(), weight_norm=True, multires=0, sphere_scale=1.0, inside_outside=False):
        super().__init__()
        self.sdf_bounding_sphere = sdf_bounding_sphere
        self.sphere_scale = sphere_scale
        self.feature_vector_size = feature_vector_size
        self.multires = multires
        self.aux_cols, self.aux_active, self.clamps = 0, False, True
        dims = [d_in] + list(dims) + [d_out + feature_vector_size]
        if multires > 0:
            dims[0] = d_in + 2 * d_in * multires
        self._make_layers(dims, geometric_init, bias, skip_in, weight_norm, multires, inside_outside)

    def forward(self, x):
        """[P, 1 + feature] : raw (unclamped) sdf in column 0 (reference network.py:79-96)."""
        P = x.shape[0]
        sdf, feat, _ = self.evaluate(x, 0, P)
        return torch.cat([sdf, feat], 1)

    def get_sdf_vals(self, x):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            P = x.shape[0]
            return self.evaluate(x, P, 0)[0]
        return self._sdf_only(x)


class ImplicitNetworkGrid(_SdfBase):
    def __init__(self, feature_vector_size, sdf_bounding_sphere, d_in, d_out, dims, geometric_init=True,
                 bias=1.0, skip_in=(), weight_norm=True, multires=0, sphere_scale=1.0, inside_outside=False,
                 base_size=16, end_size=2048, logmap=19, num_levels=16, level_dim=2, divide_factor=1.5,
                 use_grid_feature=True, debug=False):
        super().__init__()
        self.sdf_bounding_sphere = sdf_bounding_sphere
        self.sphere_scale = sphere_scale
        self.feature_vector_size = feature_vector_size
        self.multires = multires
        self.divide_factor = divide_factor
        self.grid_feature_dim = num_levels * level_dim
        self.use_grid_feature = use_grid_feature
        self.aux_cols, self.aux_active, self.clamps = self.grid_feature_dim, bool(use_grid_feature), False
        dims = [d_in] + list(dims) + [d_out + feature_vector_size]
        dims[0] += self.grid_feature_dim
        self.encoding = HashEncoder(input_dim=3, num_levels=num_levels, level_dim=level_dim, per_level_scale=2,
                                    base_resolution=base_size, log2_hashmap_size=logmap,
                                    desired_resolution=end_size)
        if multires > 0:
            dims[0] += 2 * d_in * multires
        self._make_layers(dims, geometric_init, bias, skip_in, weight_norm, multires, inside_outside)

    def forward(self, x):
        P = x.shape[0]
        sdf, feat, _ = self.evaluate(x, 0, P)
        return {'sdf': sdf, 'feature': feat}

    def get_sdf_vals(self, x):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self.evaluate(x, 0, 0)[0]
        return self._sdf_only(x)

    def mlp_parameters(self):
        params = []
        for l in self._layers():
            params += list(l.parameters())
        return params

    def grid_parameters(self):
        return self.encoding.parameters()


class RenderingNetwork(_FusedNet):
    supports_bf16x3 = True

    def __init__(self, feature_vector_size, mode, d_in, d_out, dims, weight_norm=True, multires_view=0,
                 per_image_code=False, if_hdr=False, spec=False, debug=False):
        super().__init__()
        if spec:
            raise NotImplementedError('monosdf_amd: the diffuse/specular split (spec=True) is outside the hot path')
        if mode not in ('idr', 'nerf'):
            raise NotImplementedError(mode)
        self.mode = mode
        self.spec = False
        self.feature_vector_size = feature_vector_size
        self.multires_view = multires_view
        dims = [d_in + feature_vector_size] + list(dims) + [d_out]
        if multires_view > 0:
            dims[0] += 6 * multires_view
        self.per_image_code = per_image_code
        if per_image_code:
            self.embeddings = nn.Parameter(torch.empty(1024, 32))
            self.embeddings.data.uniform_(-1e-4, 1e-4)
            dims[0] += 32
        self.num_layers = len(dims)
        self.if_hdr = if_hdr
        for l in range(self.num_layers - 1):
            setattr(self, 'lin%d' % l, WNLinear(dims[l], dims[l + 1], None, weight_norm))

    def _build_plan(self):
        shapes = [(l.out_features, l.in_features) for l in self._layers()]
        return planlib.build_color_plan(shapes, self.mode, self.multires_view, self.feature_vector_size,
                                        32 if self.per_image_code else 0, self.if_hdr)

    def forward(self, points, normals, view_dirs, feature_vectors, indices, if_pixel_input=False,
                samples_per_ray=None):
        """view_dirs: one direction per point [P,3] (reference signature) or one per ray with
        ``samples_per_ray`` given (the fused path, avoids materialising the repeat)."""
        P = points.shape[0]
        if samples_per_ray is None:
            samples_per_ray, dirs = 1, view_dirs
        else:
            dirs = view_dirs
        code = None
        if self.per_image_code:
            n_rays = P // samples_per_ray
            if not if_pixel_input:
                code = self.embeddings[indices].expand(n_rays, -1)
            else:
                code = self.embeddings[indices]
                if code.shape[0] != n_rays:      # per-point directions: expand per sample like the reference
                    code = code.unsqueeze(1).expand(-1, n_rays // code.shape[0], -1).flatten(0, 1)
            code = code.contiguous()
        fused, flat_w, flat_b, wpack, bpack = self.packed(points.device)
        rgb = ops.ColorMlpFunction.apply(points, dirs, normals, feature_vectors, code, flat_w, flat_b, wpack, bpack,
                                         fused, int(samples_per_ray), torch.is_grad_enabled())
        return {'rgb': rgb}


class MonoSDFNetwork(nn.Module):
    def __init__(self, conf, if_hdr=False):
        super().__init__()
        self.feature_vector_size = conf.get_int('feature_vector_size')
        self.scene_bounding_sphere = conf.get_float('scene_bounding_sphere', default=1.0)
        self.white_bkgd = conf.get_bool('white_bkgd', default=False)
        self.register_buffer('bg_color', torch.tensor(conf.get_list('bg_color', default=[1.0, 1.0, 1.0])).float(),
                             persistent=False)
        self.if_hdr = if_hdr
        Grid_MLP = conf.get_bool('Grid_MLP', default=False)
        self.Grid_MLP = Grid_MLP
        sphere = 0.0 if self.white_bkgd else self.scene_bounding_sphere
        cls = ImplicitNetworkGrid if Grid_MLP else ImplicitNetwork
        self.implicit_network = cls(self.feature_vector_size, sphere, **conf.get_config('implicit_network'))
        self.rendering_network = RenderingNetwork(self.feature_vector_size, **conf.get_config('rendering_network'),
                                                  if_hdr=self.if_hdr)
        self.spec = False
        self.density = LaplaceDensity(**conf.get_config('density'))
        self.ray_sampler = ErrorBoundSampler(self.scene_bounding_sphere, **conf.get_config('ray_sampler'))
        self._noise = None      # tests inject the six random draws here (SURVEY.md 8(a) RNG note)
        env = os.environ.get('MSDF_SPECULATE_ROUNDS', '1')          # see forward()
        self.speculate_rounds = False if env == '0' else ('all' if env == 'all' else True)

    def set_precision(self, precision):
        """'fp32', 'bf16x3' or 'bf16x6' matrix core for the fused MLP kernels (not a reference option)."""
        self.implicit_network.set_precision(precision)
        self.rendering_network.set_precision(precision)
        return self

    def _bg_list(self):
        bg = getattr(self, '_bg_cache', None)
        if bg is None:
            bg = [float(v) for v in self.bg_color.tolist()]
            object.__setattr__(self, '_bg_cache', bg)
        return bg

    # -- rays ------------------------------------------------------------------------------
    def _rays(self, input_dict, if_pixel_input):
        if not if_pixel_input:
            # one launch for both get_camera_params calls of the reference (network.py:505-516)
            uv, pose, intrinsics = input_dict['uv'], input_dict['pose'], input_dict['intrinsics']
            if uv.shape[0] != 1:
                raise NotImplementedError('image mode renders one image (one pose) per call, as the runner does')
            ray_dirs, ray_dirs_tmp, cam_loc = ops.camera_rays(uv[0], pose[0], intrinsics[0])
            ray_dirs, ray_dirs_tmp = ray_dirs.unsqueeze(0), ray_dirs_tmp.unsqueeze(0)
        else:
            ray_dirs = input_dict['ray_dirs'].unsqueeze(0)
            cam_loc = input_dict['ray_cam_loc']
            ray_dirs_tmp = input_dict['ray_dirs_tmp'].unsqueeze(0)
        return ray_dirs, cam_loc, ray_dirs_tmp

    def forward(self, input_dict, indices, if_pixel_input=False):
        ray_dirs, cam_loc, ray_dirs_tmp = self._rays(input_dict, if_pixel_input)
        depth_scale = ray_dirs_tmp[0, :, 2:]
        batch_size, num_pixels, _ = ray_dirs.shape
        ray_dirs = ray_dirs.reshape(-1, 3).contiguous()
        cam_loc = cam_loc.contiguous()
        device = ray_dirs.device
        noise = self._noise or {}
        net = self.implicit_network
        # |beta| + beta_min once per pass, ONE launch: the sampler and the compositor use the same value; the gradient
        # reaches density.beta through the compositor's beta_raw argument (no abs / add / sgn / mul launches of autograd)
        beta = ops.effective_beta(self.density.beta, self.density.beta_min_f)
        net.share(device)
        self.rendering_network.share(device)
        try:
            pose = input_dict['ray_pose'] if if_pixel_input else input_dict['pose'][:1]
            # The sampler's batch-global "another round?" decision (reference ray_sampler.py:179) lives on the device:
            # every launch of a round that was not asked for returns at once (its SDF evaluation included), so
            # enqueueing too many rounds costs a few empty launches and changes nothing.  The host therefore
            # enqueues as many rounds as the most demanding of the last calls needed WITHOUT reading a flag back,
            # enqueues the rest of the pass, and only then looks at the flags (on their way since the sampler
            # kernels finished).  Only if the last enqueued round asked for one more is the pass repeated, with
            # all max_total_iters rounds.  speculate_rounds: True / 'history' (this), 'all' (always all rounds:
            # no read-back at all), False (MSDF_SPECULATE_ROUNDS=0: the reference's loop, one host sync per round).
            K = self.ray_sampler.max_total_iters
            mode = self.speculate_rounds
            guess = 0 if not mode else (K if mode == 'all' else min(K, self.ray_sampler.guess_rounds()))
            for attempt in (guess, K if guess else 0):
                # the sampler's last kernel also writes the sample points and (training) the eikonal points
                z_vals, z_samples_eik, x_all = self.ray_sampler.sample(ray_dirs, cam_loc, self, speculate=attempt,
                                                                      beta0=beta)
                N, S = z_vals.shape
                P = N * S
                points_flat = x_all[:P]
                # one fused evaluation for the ray samples (clamped, with features) and the eikonal points
                sdf, feature_vectors, gradients_sdf, grad_eik = net.evaluate(x_all, P, P, save=torch.is_grad_enabled(),
                                                                                split=P)
                rgb_flat = self.rendering_network(points_flat, gradients_sdf, ray_dirs, feature_vectors, indices,
                                                  if_pixel_input=if_pixel_input, samples_per_ray=S)['rgb']
                rgb = rgb_flat.reshape(-1, S, 3)
                # the compositor also rotates the normal map into the camera frame (R^T, reference 608-616)
                weights, rgb_values, depth_values, normal_map, depth_vals = ops.CompositeFunction.apply(
                    z_vals, sdf, rgb_flat, gradients_sdf, beta, depth_scale, self.white_bkgd,
                    self._bg_list(), pose, self.density.beta, True)
                if attempt == 0 or attempt == K or self.ray_sampler.confirm():
                    break
        finally:
            net.unshare()
            self.rendering_network.unshare()

        output = {
            'rgb': rgb,
            'rgb_values': rgb_values,
            'depth_values': depth_values,
            'z_vals': z_vals,
            'depth_vals': depth_vals,
            'sdf': sdf.reshape(z_vals.shape),
            'weights': weights,
        }
        if self.training:
            output['grad_theta'], output['grad_theta_nei'] = ops.SplitRowsFunction.apply(grad_eik,
                                                                                          grad_eik.shape[0] // 2)
        output['normal_map'] = normal_map
        return output

    def volume_rendering(self, z_vals, sdf):
        """weights [N,S] (reference network.py:626-640); colours / normals are not needed for them."""
        N, S = z_vals.shape
        zeros = torch.zeros(N * S, 3, device=z_vals.device)
        ones = torch.ones(N, device=z_vals.device)
        return ops.CompositeFunction.apply(z_vals, sdf, zeros, zeros, self.density.get_beta(), ones, False,
                                           [0.0, 0.0, 0.0])[0]
