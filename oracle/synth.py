"""Numpy-seeded synthetic weights, rays and noise tensors (TEST INFRASTRUCTURE).

No dataset or checkpoint exists in the container, so every test / benchmark uses
these generators: weights follow the *distribution* of the
reference's geometric initialisation (reference: code/model/network.py:51-70 --
SDF ~ bias - |x|, a sphere of radius 0.9) but are drawn from numpy's
``default_rng`` so that fixtures only need to hold a seed.  ``jitter`` perturbs
every tensor so that no structural zero (the PE columns the geometric init
clears) hides a wrong index in a kernel.
"""
import math

import numpy as np
import torch


def _layer_dims(conf):
    ic = conf['implicit_network']
    pe = 3 + 6 * ic.get('multires', 0) if ic.get('multires', 0) > 0 else 3
    d0 = pe
    if conf.get('Grid_MLP', False):
        d0 += ic.get('num_levels', 16) * ic.get('level_dim', 2)
    dims = [d0] + list(ic['dims']) + [ic['d_out'] + conf['feature_vector_size']]
    skip = list(ic.get('skip_in', []))
    shapes = []
    for l in range(len(dims) - 1):
        out = dims[l + 1] - dims[0] if (l + 1) in skip else dims[l + 1]
        shapes.append((out, dims[l]))
    return dims, shapes, skip


def color_in_dim(conf):
    rc = conf['rendering_network']
    d = rc['d_in'] + conf['feature_vector_size']
    if rc.get('multires_view', 0) > 0:
        d += 6 * rc['multires_view']
    if rc.get('per_image_code', False):
        d += 32
    return d


def make_state(conf, seed=0, jitter=0.0, dtype=torch.float32, sdf_scale=1.0):
    """State dict (reference key names) for ``conf``; numpy default_rng(seed).

    ``sdf_scale`` multiplies the sdf row of the last SDF layer (its weight_g and bias): values below 1 make
    |sdf| under-estimate the distance to the surface, which is what keeps the error-bounded sampler from
    converging (4, 5 rounds and the exit at max_total_iters)."""
    from .hashgrid_oracle import level_geometry
    rng = np.random.default_rng(seed)
    ic = conf['implicit_network']
    dims, shapes, skip = _layer_dims(conf)
    n_lin = len(shapes)
    multires = ic.get('multires', 0)
    st = {}

    def put_linear(prefix, w, b, weight_norm=True):
        if jitter > 0:
            w = w + rng.normal(0.0, jitter * (np.abs(w).mean() + 1e-3), size=w.shape)
            b = b + rng.normal(0.0, jitter * 0.05, size=b.shape)
        if weight_norm:
            g = np.linalg.norm(w, axis=1, keepdims=True)
            if jitter > 0:
                g = g * rng.uniform(1.0 - 0.5 * jitter, 1.0 + 0.5 * jitter, size=g.shape)
            st[prefix + '.weight_g'] = g
            st[prefix + '.weight_v'] = w
        else:
            st[prefix + '.weight'] = w
        st[prefix + '.bias'] = b

    for l, (out, inn) in enumerate(shapes):
        std = math.sqrt(2) / math.sqrt(out)
        if l == n_lin - 1:
            sign = -1.0 if ic.get('inside_outside', False) else 1.0
            w = rng.normal(sign * math.sqrt(math.pi) / math.sqrt(dims[l]), 1e-4, size=(out, inn))
            b = np.full((out,), -sign * ic['bias'])
        elif multires > 0 and l == 0:
            w = np.zeros((out, inn))
            w[:, :3] = rng.normal(0.0, std, size=(out, 3))
            b = np.zeros((out,))
        elif multires > 0 and l in skip:
            w = rng.normal(0.0, std, size=(out, inn))
            w[:, -(dims[0] - 3):] = 0.0
            b = np.zeros((out,))
        else:
            w = rng.normal(0.0, std, size=(out, inn))
            b = np.zeros((out,))
        put_linear('implicit_network.lin%d' % l, w, b, ic.get('weight_norm', True))

    if conf.get('Grid_MLP', False):
        geo = level_geometry(ic)
        amp = 1e-4 if jitter == 0 else 0.05
        st['implicit_network.encoding.embeddings'] = rng.uniform(-amp, amp, size=(geo['n_entries'], geo['C']))
        st['implicit_network.encoding.offsets'] = np.asarray(geo['offsets'], dtype=np.int32)

    rc = conf['rendering_network']
    cdims = [color_in_dim(conf)] + list(rc['dims']) + [rc['d_out']]
    for l in range(len(cdims) - 1):
        bound = 1.0 / math.sqrt(cdims[l])
        w = rng.uniform(-bound, bound, size=(cdims[l + 1], cdims[l]))
        b = rng.uniform(-bound, bound, size=(cdims[l + 1],))
        put_linear('rendering_network.lin%d' % l, w, b, rc.get('weight_norm', True))
    if rc.get('per_image_code', False):
        st['rendering_network.embeddings'] = rng.uniform(-1e-4, 1e-4, size=(1024, 32))
    st['density.beta'] = np.asarray(conf['density']['params_init']['beta'], dtype=np.float64)

    if sdf_scale != 1.0:
        last = 'implicit_network.lin%d' % (n_lin - 1)
        st[last + '.bias'] = st[last + '.bias'].copy()
        st[last + '.bias'][0] *= sdf_scale
        key = last + ('.weight_g' if ic.get('weight_norm', True) else '.weight')
        st[key] = st[key].copy()
        st[key][0] *= sdf_scale

    out = {}
    for k, v in st.items():
        t = torch.from_numpy(np.ascontiguousarray(v))
        out[k] = t if t.dtype in (torch.int32, torch.int64) else t.to(dtype)
    return out


def make_rays(n_rays, seed=1, dtype=torch.float32, random_pose=False):
    """Pixel-mode input dict: origins U(-0.2,0.2)^3 inside the sphere, unit directions."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-0.2, 0.2, size=(n_rays, 3))
    d = rng.normal(size=(n_rays, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pose = np.tile(np.eye(4), (n_rays, 1, 1))
    d_cam = d.copy()
    if random_pose:
        q, _ = np.linalg.qr(rng.normal(size=(n_rays, 3, 3)))
        pose[:, :3, :3] = q
        pose[:, :3, 3] = o
        d_cam = np.einsum('nji,nj->ni', q, d)          # R^T d
        d_cam = d_cam / np.maximum(np.abs(d_cam[:, 2:3]), 0.2)   # un-normalised, z ~ 1
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    return {'ray_dirs': t(d), 'ray_cam_loc': t(o), 'ray_dirs_tmp': t(d_cam), 'ray_pose': t(pose)}


def make_noise(conf, n_rays, n_dense, seed=2, dtype=torch.float32):
    """The six training-mode random draws as explicit tensors (SURVEY.md 8(a) RNG note).

    ``n_dense`` is the width of the sampler's dense z array when the loop ends
    (128*rounds); 'extra_idx' indexes into it like randperm(n_dense)[:32].
    """
    rng = np.random.default_rng(seed)
    sc = conf['ray_sampler']
    r = float(conf.get('scene_bounding_sphere', 1.0))
    n_out = sc['N_samples'] + sc['N_samples_extra'] + 2
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    return {
        'jitter': t(rng.uniform(size=(n_rays, sc['N_samples_eval']))),
        'final_u': t(rng.uniform(size=(n_rays, sc['N_samples']))),
        'extra_idx': torch.from_numpy(rng.permutation(n_dense)[:sc['N_samples_extra']].astype(np.int64)),
        'eik_idx': torch.from_numpy(rng.integers(0, n_out, size=(n_rays,)).astype(np.int64)),
        'eik_uniform': t(rng.uniform(-r, r, size=(n_rays, 3))),
        'nei_rand': t(rng.uniform(size=(2 * n_rays, 3))),
    }


def make_noise_table(conf, n_rays, seed=2, dtype=torch.float32):
    """make_noise for a call whose number of sampler rounds is not known in advance: 'extra_idx' is a table
    [max_total_iters, N_samples_extra] whose row k-1 holds the columns for a dense set of 128 k samples."""
    sc = conf['ray_sampler']
    noise = make_noise(conf, n_rays, sc['N_samples_eval'], seed, dtype)
    rows = []
    for k in range(sc['max_total_iters']):
        rng = np.random.default_rng(seed * 131 + 7 * k + 1)
        rows.append(rng.permutation(sc['N_samples_eval'] * (k + 1))[:sc['N_samples_extra']].astype(np.int64))
    noise['extra_idx'] = torch.from_numpy(np.stack(rows))
    return noise


def analytic_targets(rays, radius=0.6):
    """Supervision for the training-trajectory fixture, a closed-form scene so that nothing but seeds has to be
    stored: the rays (origins inside) hit the inside of a sphere of `radius`; colour is a smooth function of the hit
    point, the depth cue is the hit distance up to the monocular scale, the normal cue the sphere's inward normal
    in the camera frame.  Shapes as the reference's data loader delivers them ([1, N, C])."""
    o, d = rays['ray_cam_loc'].double(), rays['ray_dirs'].double()
    b = (o * d).sum(-1)
    c = (o * o).sum(-1) - radius * radius
    t = -b + torch.sqrt(b * b - c)
    p = o + t.unsqueeze(-1) * d
    rgb = 0.5 + 0.4 * torch.sin(3.0 * p + torch.tensor([0.0, 1.0, 2.0], dtype=torch.float64))
    n_world = -p / radius
    rot = rays['ray_pose'][:, :3, :3].double().transpose(1, 2)
    n_cam = (rot @ n_world.unsqueeze(-1)).squeeze(-1)
    depth = t * rays['ray_dirs_tmp'][:, 2].double().abs()
    f = lambda a: a.float()[None]
    return {'rgb': f(rgb), 'depth': f((depth / 50.0).unsqueeze(-1)), 'normal': f(n_cam),
            'mask': torch.ones(1, o.shape[0], 1)}


def table_fingerprint(grad, offsets, n_proj=4, n_top=64, seed=2024):
    """Fingerprint of a hash-grid table gradient [n_entries, C] that is too large to commit (48.8 MB) and too sparse
    for a 16-entry sample to say anything: per-level sum of |g| and of g^2 (which LEVELS received what), projections
    onto fixed +-1 vectors (every entry's position counts: a contribution scattered to a wrong entry changes them),
    and the n_top largest entries with their indices.  Everything in float64."""
    g = np.asarray(grad.detach().cpu().double().numpy() if torch.is_tensor(grad) else grad, dtype=np.float64)
    off = [int(o) for o in np.asarray(offsets).reshape(-1)]
    level_abs = np.array([np.abs(g[a:b]).sum() for a, b in zip(off[:-1], off[1:])])
    level_sq = np.array([(g[a:b] ** 2).sum() for a, b in zip(off[:-1], off[1:])])
    rng = np.random.default_rng(seed)
    flat = g.reshape(-1)
    proj = np.array([(flat * (rng.integers(0, 2, size=flat.size, dtype=np.int8) * 2 - 1)).sum() for _ in range(n_proj)])
    top = np.argsort(-np.abs(flat), kind='stable')[:n_top]
    return {'level_abs': level_abs, 'level_sq': level_sq, 'proj': proj, 'top_idx': top.astype(np.int64),
            'top_val': flat[top]}
