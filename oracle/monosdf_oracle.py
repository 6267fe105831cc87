"""CPU restatement of the MonoSDF volume-rendering path (torch, CPU, fp32 or fp64).

TEST INFRASTRUCTURE (see oracle/__init__.py) -- the product never imports this.

It is a functional restatement: parameters come in as a flat ``state`` dict whose
keys are the reference's state-dict keys (checkpoint row),
every random draw is an explicit input (``noise``), and nothing touches a GPU.
Each function cites the reference lines it follows (paths relative to
/root/reference/code).  Pinned against the real reference by
``oracle/make_golden.py`` -> ``tests/golden/*.npz`` -> ``tests/test_oracle_golden.py``.
"""
import math

import torch

from . import hashgrid_oracle


# ----------------------------------------------------------------------------
# small pieces
# ----------------------------------------------------------------------------
def positional_encoding(x, n_freqs):
    """[x, sin(2^k x), cos(2^k x)]_{k<n_freqs}  (model/embedder.py:10-36,38-50)."""
    if n_freqs <= 0:
        return x
    cols = [x]
    for k in range(n_freqs):
        f = float(2.0 ** k)
        cols.append(torch.sin(x * f))
        cols.append(torch.cos(x * f))
    return torch.cat(cols, dim=-1)


def effective_weight(state, prefix):
    """Old-style weight_norm: w = g * v / ||v||, norm per output row (network.py:72-73)."""
    if prefix + '.weight_g' in state:
        # torch._weight_norm is the primitive nn.utils.weight_norm's hook calls (same rounding)
        return torch._weight_norm(state[prefix + '.weight_v'], state[prefix + '.weight_g'], 0)
    return state[prefix + '.weight']


def softplus100(a):
    """nn.Softplus(beta=100): linear above beta*a > 20 (network.py:77)."""
    return torch.nn.functional.softplus(a, beta=100.0, threshold=20.0)


def laplace_density(sdf, beta):
    """sigma = (1/beta) (1/2 + 1/2 sign(s) expm1(-|s|/beta))  (model/density.py:21-26)."""
    return (1.0 / beta) * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))


def get_beta(state, conf):
    """|beta| + beta_min (model/density.py:28-30)."""
    return state['density.beta'].abs() + conf['density']['beta_min']


# ----------------------------------------------------------------------------
# SDF network (both classes)
# ----------------------------------------------------------------------------
def _n_sdf_layers(conf):
    return len(conf['implicit_network']['dims']) + 1


def sdf_network_raw(state, conf, x):
    """Forward of ImplicitNetwork / ImplicitNetworkGrid -> [P, 1+feature] (network.py:79-96,247-275)."""
    ic = conf['implicit_network']
    n_lin = _n_sdf_layers(conf)
    skip = list(ic.get('skip_in', []))
    inp = positional_encoding(x, ic.get('multires', 0))
    if conf.get('Grid_MLP', False):
        if ic.get('use_grid_feature', True):
            feat = hashgrid_oracle.hash_encode_autograd(
                x / ic.get('divide_factor', 1.5), state['implicit_network.encoding.embeddings'],
                hashgrid_oracle.level_geometry(ic))
        else:
            width = ic.get('num_levels', 16) * ic.get('level_dim', 2)
            feat = torch.zeros(x.shape[0], width, dtype=x.dtype)
        inp = torch.cat([inp, feat], dim=-1)
    h = inp
    for l in range(n_lin):
        if l in skip:
            h = torch.cat([h, inp], dim=1) / math.sqrt(2.0)
        w = effective_weight(state, 'implicit_network.lin%d' % l)
        h = torch.nn.functional.linear(h, w, state['implicit_network.lin%d.bias' % l])
        if l < n_lin - 1:
            h = softplus100(h)
    return h


def _sphere_radius(conf):
    """0 disables the clamp: white_bkgd, or the grid class which never clamps (network.py:490,290-309)."""
    if conf.get('white_bkgd', False) or conf.get('Grid_MLP', False):
        return 0.0
    return float(conf.get('scene_bounding_sphere', 1.0))


def _clamp_sdf(conf, sdf, x):
    r = _sphere_radius(conf)
    if r > 0.0:
        scale = conf['implicit_network'].get('sphere_scale', 1.0)
        sdf = torch.minimum(sdf, scale * (r - x.norm(2, 1, keepdim=True)))
    return sdf


def get_sdf_vals(state, conf, x):
    """network.py:131-137 (clamped) / 307-309 (grid: raw)."""
    return _clamp_sdf(conf, sdf_network_raw(state, conf, x)[:, :1], x)


def get_outputs(state, conf, x, create_graph=True):
    """sdf (clamped), feature vectors, d sdf/dx  (network.py:111-129, 290-305)."""
    x = x.detach().requires_grad_(True)
    out = sdf_network_raw(state, conf, x)
    sdf = _clamp_sdf(conf, out[:, :1], x)
    grad = torch.autograd.grad(sdf, x, torch.ones_like(sdf), create_graph=create_graph,
                               retain_graph=True)[0]
    return sdf, out[:, 1:], grad


def gradient_sdf(state, conf, x, create_graph=True):
    """Gradient of the UNCLAMPED sdf (network.py:98-109, 277-288)."""
    x = x.detach().requires_grad_(True)
    y = sdf_network_raw(state, conf, x)[:, :1]
    return torch.autograd.grad(y, x, torch.ones_like(y), create_graph=create_graph,
                               retain_graph=True)[0]


# ----------------------------------------------------------------------------
# colour network
# ----------------------------------------------------------------------------
def color_network(state, conf, points, normals, view_dirs, feats, indices=None,
                  if_pixel_input=False, if_hdr=False):
    """RenderingNetwork.forward, mode idr/nerf, no spec branch (network.py:389-470)."""
    rc = conf['rendering_network']
    v = positional_encoding(view_dirs, rc.get('multires_view', 0))
    if rc['mode'] == 'idr':
        h = torch.cat([points, v, normals, feats], dim=-1)
    elif rc['mode'] == 'nerf':
        h = torch.cat([v, feats], dim=-1)
    else:
        raise NotImplementedError(rc['mode'])
    if rc.get('per_image_code', False):
        emb = state['rendering_network.embeddings']
        if not if_pixel_input:
            code = emb[indices].expand(h.shape[0], -1)
        else:
            n_s = h.shape[0] // indices.shape[0]
            code = emb[indices].unsqueeze(1).expand(-1, n_s, -1).flatten(0, 1)
        h = torch.cat([h, code], dim=-1)
    n_lin = len(rc['dims']) + 1
    for l in range(n_lin):
        w = effective_weight(state, 'rendering_network.lin%d' % l)
        h = torch.nn.functional.linear(h, w, state['rendering_network.lin%d.bias' % l])
        if l < n_lin - 1:
            h = torch.relu(h)
    return torch.relu(h) if if_hdr else torch.sigmoid(h)


# ----------------------------------------------------------------------------
# volume rendering
# ----------------------------------------------------------------------------
def transmittance_weights(z_vals, density):
    """alpha_i * T_i with an exclusive prefix sum of free energy (network.py:626-640)."""
    dists = z_vals[:, 1:] - z_vals[:, :-1]
    dists = torch.cat([dists, torch.full_like(dists[:, :1], 1e10)], -1)
    free = dists * density
    shifted = torch.cat([torch.zeros_like(free[:, :1]), free[:, :-1]], -1)
    alpha = 1 - torch.exp(-free)
    trans = torch.exp(-torch.cumsum(shifted, dim=-1))
    return alpha * trans, trans, dists


def volume_rendering(state, conf, z_vals, sdf):
    """MonoSDFNetwork.volume_rendering (network.py:626-640)."""
    dens = laplace_density(sdf, get_beta(state, conf)).reshape(-1, z_vals.shape[1])
    return transmittance_weights(z_vals, dens)[0]


# ----------------------------------------------------------------------------
# samplers
# ----------------------------------------------------------------------------
def far_from_cube(origins, dirs, bound, near_clip, far_clip):
    """Ray/AABB slab test (model/ray_sampler.py:48-60)."""
    t0 = (-bound - origins) / (dirs + 1e-15)
    t1 = (bound - origins) / (dirs + 1e-15)
    near = torch.minimum(t0, t1).max(dim=-1, keepdim=True)[0]
    far = torch.maximum(t0, t1).min(dim=-1, keepdim=True)[0]
    miss = far < near
    near = torch.where(miss, torch.full_like(near, 1e9), near)
    far = torch.where(miss, torch.full_like(far, 1e9), far)
    return near.clamp(min=near_clip), far.clamp(max=far_clip)


def sampler_far(conf):
    """far = 2 R 1.75 (model/ray_sampler.py:19,91)."""
    return 2.0 * float(conf.get('scene_bounding_sphere', 1.0)) * 1.75


def uniform_z(conf, ray_dirs, cam_loc, n_samples, jitter=None):
    """UniformSampler.get_z_vals with take_sphere_intersection=True (ray_sampler.py:63-83)."""
    sc = conf['ray_sampler']
    r = float(conf.get('scene_bounding_sphere', 1.0))
    _, far = far_from_cube(cam_loc, ray_dirs, r, sc['near'], sampler_far(conf))
    near = torch.full_like(far, sc['near'])
    t = torch.linspace(0., 1., steps=n_samples, dtype=ray_dirs.dtype)
    z = near * (1. - t) + far * t
    if jitter is not None:
        mids = .5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * jitter
    return z, near, far


def error_bound(beta, sdf, dists, d_star):
    """Opacity error bound for one beta per ray (ray_sampler.py:264-272)."""
    dens = laplace_density(sdf, beta)
    shifted = torch.cat([torch.zeros_like(dists[:, :1]), dists * dens[:, :-1]], dim=-1)
    integral = torch.cumsum(shifted, dim=-1)
    per_section = torch.exp(-d_star / beta) * (dists ** 2.) / (4 * beta ** 2)
    err_int = torch.cumsum(per_section, dim=-1)
    bound = (torch.clamp(torch.exp(err_int), max=1.e6) - 1.0) * torch.exp(-integral[:, :-1])
    return bound.max(-1)[0]


def _d_star(z_vals, d):
    """Triangle bound on the minimal |sdf| inside each interval (ray_sampler.py:141-153)."""
    a = z_vals[:, 1:] - z_vals[:, :-1]
    b, c = d[:, :-1].abs(), d[:, 1:].abs()
    first = a.pow(2) + b.pow(2) <= c.pow(2)
    second = a.pow(2) + c.pow(2) <= b.pow(2)
    s = (a + b + c) / 2.0
    area = s * (s - a) * (s - b) * (s - c)
    tri = ~first & ~second & (b + c - a > 0)
    zero = torch.zeros_like(a)
    out = torch.where(first, b, zero)
    out = torch.where(second, c, out)
    out = torch.where(tri, 2.0 * torch.sqrt(torch.where(tri, area, zero)) / a, out)
    same_sign = (d[:, 1:].sign() * d[:, :-1].sign() == 1)
    return same_sign.to(a.dtype) * out, a


def _inverse_cdf(cdf, bins, u):
    """searchsorted(right=True) + linear interpolation (ray_sampler.py:216-228)."""
    inds = torch.searchsorted(cdf.contiguous(), u.contiguous(), right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_b, bin_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return bin_b + (u - cdf_b) / denom * (bin_a - bin_b)


def error_bound_sampler(state, conf, ray_dirs, cam_loc, training=False, noise=None,
                        sdf_fn=None, trace=None, max_reduce=None):
    """ErrorBoundSampler.get_z_vals (ray_sampler.py:110-262), inverse_sphere_bg=False.

    noise (training only): 'jitter' [N,128], 'final_u' [N,64], 'extra_idx' [32] (int64),
    'eik_idx' [N] (int64).  In eval mode 'eik_idx' defaults to zeros (unused by outputs).
    ``sdf_fn(points)->[P,1]`` overrides the network (used to test the sampler alone).
    ``max_reduce(t)->t`` (multi-rank tests): applied to the batch's max beta before the convergence test, e.g. an
    all-reduce(MAX), so that a shard runs the rounds the whole batch would.
    ``trace`` (dict) receives the number of rounds, the final beta and, under 'per_round', each round's
    sorted z, merged sdf, d*, beta after the bisection, cdf, u and new samples.
    """
    sc = conf['ray_sampler']
    noise = noise or {}
    dt = ray_dirs.dtype
    n_rays = ray_dirs.shape[0]
    eps, n_eval, n_final = sc['eps'], sc['N_samples_eval'], sc['N_samples']
    add_tiny = sc.get('add_tiny', 1.0e-6)
    if sdf_fn is None:
        def sdf_fn(p):
            with torch.no_grad():
                return get_sdf_vals(state, conf, p)
    beta0 = get_beta(state, conf).detach().to(dt)

    z_vals, _, _ = uniform_z(conf, ray_dirs, cam_loc, n_eval,
                             noise.get('jitter') if training else None)
    samples, order = z_vals, None
    dists = z_vals[:, 1:] - z_vals[:, :-1]
    # Lemma-2 upper bound; the constant is formed in the working dtype like the reference does
    lemma = 1.0 / (4.0 * torch.log(torch.tensor(eps + 1.0, dtype=dt)))
    beta = torch.sqrt(lemma * (dists ** 2.).sum(-1))
    rounds, unconverged, sdf = 0, True, None
    while unconverged and rounds < sc['max_total_iters']:
        pts = (cam_loc.unsqueeze(1) + samples.unsqueeze(2) * ray_dirs.unsqueeze(1)).reshape(-1, 3)
        new_sdf = sdf_fn(pts).reshape(n_rays, -1).detach()
        if order is not None:
            sdf = torch.gather(torch.cat([sdf, new_sdf], -1), 1, order)
        else:
            sdf = new_sdf
        d_star, dists = _d_star(z_vals, sdf)

        # bisection on beta per ray (ray_sampler.py:157-165)
        err = error_bound(beta0, sdf, dists, d_star)
        beta = torch.where(err <= eps, beta0.expand_as(beta), beta)
        lo, hi = beta0.expand(n_rays).clone(), beta.clone()
        for _ in range(sc['beta_iters']):
            mid = (lo + hi) / 2.
            err = error_bound(mid.unsqueeze(-1), sdf, dists, d_star)
            hi = torch.where(err <= eps, mid, hi)
            lo = torch.where(err > eps, mid, lo)
        beta = hi

        dens = laplace_density(sdf, beta.unsqueeze(-1))
        weights, trans, dists_full = transmittance_weights(z_vals, dens)
        rounds += 1
        bmax = beta.max() if max_reduce is None else max_reduce(beta.max())
        unconverged = bool(bmax > beta0)
        more = unconverged and rounds < sc['max_total_iters']
        if more:
            # sample proportionally to the current error bound (ray_sampler.py:181-194)
            n_new = n_eval
            b = beta.unsqueeze(-1)
            per_section = torch.exp(-d_star / b) * (dists ** 2.) / (4 * b ** 2)
            err_int = torch.cumsum(per_section, dim=-1)
            pdf = (torch.clamp(torch.exp(err_int), max=1.e6) - 1.0) * trans[:, :-1] + add_tiny
        else:
            n_new = n_final
            pdf = weights[..., :-1] + 1e-5
        pdf = pdf / torch.sum(pdf, -1, keepdim=True)
        cdf = torch.cumsum(pdf, -1)
        cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
        if more or not training:
            u = torch.linspace(0., 1., steps=n_new, dtype=dt).unsqueeze(0).repeat(n_rays, 1)
        else:
            u = noise['final_u'].to(dt)
        samples = _inverse_cdf(cdf, z_vals, u)
        if trace is not None:
            trace.setdefault('per_round', []).append(
                {'z': z_vals, 'sdf': sdf, 'dstar': d_star, 'beta': beta, 'cdf': cdf, 'u': u, 'samples': samples})
        if more:
            z_vals, order = torch.sort(torch.cat([z_vals, samples], -1), -1)

    if trace is not None:
        trace['rounds'] = rounds
        trace['beta'] = beta
        trace['z_dense'] = z_vals
    near = torch.full((n_rays, 1), sc['near'], dtype=dt)
    far = torch.full((n_rays, 1), sampler_far(conf), dtype=dt)
    n_extra = sc['N_samples_extra']
    if n_extra > 0:
        if training:
            pick = noise['extra_idx']
        else:
            pick = torch.linspace(0, z_vals.shape[1] - 1, n_extra).long()
        extra = torch.cat([near, far, z_vals[:, pick]], -1)
    else:
        extra = torch.cat([near, far], -1)
    z_out, _ = torch.sort(torch.cat([samples, extra], -1), -1)
    eik_idx = noise.get('eik_idx')
    if eik_idx is None:
        eik_idx = torch.zeros(n_rays, dtype=torch.long)
    z_eik = torch.gather(z_out, 1, eik_idx.unsqueeze(-1))
    return z_out, z_eik


# ----------------------------------------------------------------------------
# ray generation (the step before the path)
# ----------------------------------------------------------------------------
def camera_rays(uv, pose, intrinsics):
    """get_camera_params + lift for 4x4 pose matrices (utils/rend_util.py:63-91,105-118)."""
    cam_loc = pose[:, :3, 3]
    fx, fy = intrinsics[:, 0, 0:1], intrinsics[:, 1, 1:2]
    cx, cy, sk = intrinsics[:, 0, 2:3], intrinsics[:, 1, 2:3], intrinsics[:, 0, 1:2]
    x, y = uv[:, :, 0], uv[:, :, 1]
    z = torch.ones_like(x)
    x_l = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    y_l = (y - cy) / fy * z
    cam_pts = torch.stack((x_l, y_l, z, torch.ones_like(z)), dim=-1)
    world = torch.bmm(pose, cam_pts.permute(0, 2, 1)).permute(0, 2, 1)[:, :, :3]
    dirs = torch.nn.functional.normalize(world - cam_loc[:, None, :], dim=2)
    return dirs, cam_loc


# ----------------------------------------------------------------------------
# the whole forward
# ----------------------------------------------------------------------------
def render(state, conf, inputs, indices=None, if_pixel_input=False, training=False,
           noise=None, if_hdr=False, z_override=None):
    """MonoSDFNetwork.forward (network.py:502-624), spec branch excluded.

    noise (training): sampler keys (see error_bound_sampler) + 'eik_uniform' [N,3] in
    [-R,R], 'nei_rand' [2N,3] in [0,1).  ``z_override=(z_vals, z_eik)`` skips the sampler.
    """
    noise = noise or {}
    if not if_pixel_input:
        ray_dirs, cam_loc = camera_rays(inputs['uv'], inputs['pose'], inputs['intrinsics'])
        eye = torch.eye(4, dtype=inputs['pose'].dtype)[None]
        dirs_cam, _ = camera_rays(inputs['uv'], eye, inputs['intrinsics'])
        cam_loc = cam_loc.unsqueeze(1).repeat(1, ray_dirs.shape[1], 1).reshape(-1, 3)
    else:
        ray_dirs = inputs['ray_dirs'].unsqueeze(0)
        cam_loc = inputs['ray_cam_loc']
        dirs_cam = inputs['ray_dirs_tmp'].unsqueeze(0)
    depth_scale = dirs_cam[0, :, 2:]
    n_batch, n_pix, _ = ray_dirs.shape
    ray_dirs = ray_dirs.reshape(-1, 3)

    if z_override is not None:
        z_vals, z_eik = z_override
    else:
        z_vals, z_eik = error_bound_sampler(state, conf, ray_dirs, cam_loc, training, noise)
    n_s = z_vals.shape[1]
    pts = (cam_loc.unsqueeze(1) + z_vals.unsqueeze(2) * ray_dirs.unsqueeze(1)).reshape(-1, 3)
    dirs = ray_dirs.unsqueeze(1).repeat(1, n_s, 1).reshape(-1, 3)

    sdf, feats, grads = get_outputs(state, conf, pts)
    rgb = color_network(state, conf, pts, grads, dirs, feats, indices, if_pixel_input,
                        if_hdr).reshape(-1, n_s, 3)
    weights = volume_rendering(state, conf, z_vals, sdf)
    rgb_values = torch.sum(weights.unsqueeze(-1) * rgb, 1)
    depth = torch.sum(weights * z_vals, 1, keepdim=True) / (weights.sum(dim=1, keepdim=True) + 1e-8)
    depth = depth_scale * depth
    if conf.get('white_bkgd', False):
        bg = torch.tensor(conf.get('bg_color', [1.0, 1.0, 1.0]), dtype=rgb.dtype)
        rgb_values = rgb_values + (1. - weights.sum(-1, keepdim=True)) * bg.unsqueeze(0)
    out = {'rgb': rgb, 'rgb_values': rgb_values, 'depth_values': depth, 'z_vals': z_vals,
           'depth_vals': z_vals * depth_scale, 'sdf': sdf.reshape(z_vals.shape), 'weights': weights}

    if training:
        eik = torch.cat([noise['eik_uniform'],
                         (cam_loc.unsqueeze(1) + z_eik.unsqueeze(2) * ray_dirs.unsqueeze(1)).reshape(-1, 3)], 0)
        eik = torch.cat([eik, eik + (noise['nei_rand'] - 0.5) * 0.01], 0)
        g = gradient_sdf(state, conf, eik)
        out['grad_theta'] = g[:g.shape[0] // 2]
        out['grad_theta_nei'] = g[g.shape[0] // 2:]

    normals = grads / (grads.norm(2, -1, keepdim=True) + 1e-6)
    normal_map = torch.sum(weights.unsqueeze(-1) * normals.reshape(-1, n_s, 3), 1)
    if if_pixel_input:
        rot = inputs['ray_pose'][:, :3, :3].transpose(1, 2)
        normal_map = (rot @ normal_map.unsqueeze(-1)).squeeze(-1)
    else:
        rot = inputs['pose'][0, :3, :3].permute(1, 0)
        normal_map = (rot @ normal_map.permute(1, 0)).permute(1, 0)
    out['normal_map'] = normal_map
    return out


def render_uniform(state, conf, inputs, indices, n_samples):
    """BASELINE.json configs[0]: the same path with UniformSampler(N_samples=n_samples, take_sphere_intersection=True)
    in place of the error-bounded sampler (ray_sampler.py:16-83), eval mode, pixel-mode inputs."""
    z, _, _ = uniform_z(conf, inputs['ray_dirs'], inputs['ray_cam_loc'], n_samples)
    return render(state, conf, inputs, indices, True, False, None, z_override=(z, None))


def probe_loss(out):
    """The fixed scalar used for timing / gradient parity:
    mean|rgb| + 0.05 eikonal + 0.05 mean|normal| + 0.1 mean depth (+0.005 smooth)."""
    loss = out['rgb_values'].abs().mean() + 0.05 * out['normal_map'].abs().mean() \
        + 0.1 * out['depth_values'].mean()
    if 'grad_theta' in out:
        g1, g2 = out['grad_theta'], out['grad_theta_nei']
        loss = loss + 0.05 * ((g1.norm(2, dim=1) - 1) ** 2).mean()
        n1 = g1 / (g1.norm(2, dim=1).unsqueeze(-1) + 1e-5)
        n2 = g2 / (g2.norm(2, dim=1).unsqueeze(-1) + 1e-5)
        loss = loss + 0.005 * torch.norm(n1 - n2, dim=-1).mean()
    return loss


# ----------------------------------------------------------------------------
# inference drivers -- restated for parity tests
# ----------------------------------------------------------------------------
def render_image(state, conf, inputs, indices, total_pixels, split_n_pixels=1024,
                 keys=('rgb_values', 'normal_map', 'depth_values')):
    """Chunked eval render (utils/general.py:28-58; evaluation/eval.py:105-120)."""
    outs = []
    for idx in torch.split(torch.arange(total_pixels), split_n_pixels, dim=0):
        chunk = dict(inputs)
        chunk['uv'] = torch.index_select(inputs['uv'], 1, idx)
        o = render(state, conf, chunk, indices, False, False, None)
        outs.append({k: o[k].detach() for k in keys})
    return {k: torch.cat([o[k] for o in outs], 0) for k in keys}


def sdf_volume_block(sdf_fn, mins, maxs, cropN):
    """One block of plots.get_surface_sliding's coarse-to-fine SDF evaluation (utils/plots.py:135-190)."""
    import numpy as np
    axes = [torch.tensor(np.linspace(mins[d], maxs[d], cropN)) for d in range(3)]
    xx, yy, zz = torch.meshgrid(*axes, indexing='ij')
    pts = torch.vstack([xx.flatten(), yy.flatten(), zz.flatten()]).T.float()
    pool = torch.nn.AvgPool3d(2, stride=2)
    up = torch.nn.Upsample(scale_factor=2, mode='nearest')
    p = pts.reshape(cropN, cropN, cropN, 3).permute(3, 0, 1, 2)
    pyr = [p]
    for _ in range(3):
        p = pool(p[None])[0]
        pyr.append(p)
    pyr = pyr[::-1]
    mask, vals = None, None
    threshold = 2 * (maxs[0] - mins[0]) / cropN * 8
    for pid, q in enumerate(pyr):
        n = q.shape[-1]
        flat = q.reshape(3, -1).permute(1, 0).contiguous()
        if mask is None:
            vals = sdf_fn(flat).reshape(-1)
        else:
            m = mask.reshape(-1)
            if bool(m.any()):
                vals[m] = sdf_fn(flat[m].contiguous()).reshape(-1)
        if pid < 3:
            mask = up((vals.abs() < threshold).reshape(n, n, n)[None, None].float()).bool()
            vals = up(vals.reshape(n, n, n)[None, None]).reshape(-1)
        threshold /= 2.
    return vals.reshape(cropN, cropN, cropN)
