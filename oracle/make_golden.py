"""Generate tests/golden/*.npz from the REAL reference (container-only).

TEST INFRASTRUCTURE.  Run as ``python -m oracle.make_golden`` from the repo root in
the authoring container (where /root/reference exists).  It imports the reference's
Python model classes on CPU through ``oracle/ref_loader.py`` and stores, per case:
the generator seeds (weights/rays are re-created from ``oracle/synth.py`` by the
tests, so no weight blobs are committed), the random draws the reference made, and
the reference's outputs / parameter gradients.  Fixtures hold data only.

Hash-grid cases run the reference's Python wiring over the *restated* kernels
(ref_loader.FakeHashBackend) -- they pin the wiring, not the kernel arithmetic.
"""
import os
import sys
import zlib

import numpy as np
import torch

from . import config, monosdf_oracle as mo, ref_loader, synth

OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def conf_from_spec(spec):
    kind = spec['kind']
    if kind == 'mlp':
        c = config.mlp_config(spec['width'], spec.get('depth', 8), spec.get('beta', 0.1))
    elif kind == 'gridless':
        c = config.gridless_config(spec['width'], spec.get('depth', 8), spec.get('beta', 0.1))
    elif kind == 'grid':
        c = config.grid_config(spec['width'], spec.get('beta', 0.1), spec.get('num_levels', 16),
                               spec.get('level_dim', 2), spec.get('logmap', 19),
                               spec.get('base_size', 16), spec.get('end_size', 2048))
    else:
        raise ValueError(kind)
    if spec.get('white_bkgd', False):
        c['white_bkgd'] = True
    if spec.get('per_image_code', False):
        c['rendering_network']['per_image_code'] = True
    if spec.get('render_mode', 'idr') == 'nerf':
        # view directions + features only (reference network.py:437-440); no conf of the reference uses it
        c['rendering_network']['mode'] = 'nerf'
        c['rendering_network']['d_in'] = 3
    return c


def make_inputs(spec):
    n = spec['n_rays']
    if spec.get('image_mode', False):
        rng = np.random.default_rng(spec.get('ray_seed', 1))
        uv = torch.from_numpy(rng.uniform(0, 384, size=(1, n, 2))).float()
        intr = torch.eye(4)[None].clone()
        intr[0, 0, 0], intr[0, 1, 1], intr[0, 0, 2], intr[0, 1, 2], intr[0, 0, 1] = 300., 310., 192., 190., 0.5
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        pose = torch.eye(4)[None].clone()
        pose[0, :3, :3] = torch.from_numpy(q).float()
        pose[0, :3, 3] = torch.tensor([0.1, -0.15, 0.05])
        return {'uv': uv, 'pose': pose, 'intrinsics': intr}, torch.tensor([3])
    rays = synth.make_rays(n, seed=spec.get('ray_seed', 1), random_pose=True)
    return rays, torch.arange(n) % 7


CASES = {
    # name: spec
    'mlp_w64_eval': dict(kind='mlp', width=64, n_rays=24, jitter=0.3, training=False),
    'mlp_w64_eval_sharp': dict(kind='mlp', width=64, n_rays=24, jitter=0.0, beta=0.01, training=False),
    'mlp_w64_eval_vsharp': dict(kind='mlp', width=64, n_rays=16, jitter=0.05, beta=0.002, training=False),
    'mlp_w64_eval_k3': dict(kind='mlp', width=64, n_rays=8, jitter=0.02, beta=0.0003, training=False),
    # sdf_scale < 1: |sdf| under-estimates the distance to the surface, the sampler needs 4 / 5 rounds or leaves
    # the loop at max_total_iters without converging (ray_sampler.py:125,179-207); trace = per-round intermediates
    'mlp_w64_eval_k4': dict(kind='mlp', width=64, n_rays=6, jitter=0.05, beta=0.0005, sdf_scale=0.25,
                            training=False, trace=True),
    'mlp_w64_eval_k5': dict(kind='mlp', width=64, n_rays=6, jitter=0.3, beta=0.002, sdf_scale=0.25,
                            training=False, trace=True),
    'mlp_w64_eval_k5nc': dict(kind='mlp', width=64, n_rays=6, jitter=0.3, beta=0.001, sdf_scale=0.25,
                              training=False, trace=True),
    'mlp_w64_train_k4': dict(kind='mlp', width=64, n_rays=6, jitter=0.05, beta=0.0005, sdf_scale=0.25,
                             training=True, grads=True, trace=True),
    'mlp_w64_train_k5nc': dict(kind='mlp', width=64, n_rays=6, jitter=0.3, beta=0.001, sdf_scale=0.25,
                               training=True, grads=True, trace=True),
    'mlp_w64_eval_k2_trace': dict(kind='mlp', width=64, n_rays=6, jitter=0.0, beta=0.01, training=False, trace=True),
    'mlp_w64_train': dict(kind='mlp', width=64, n_rays=24, jitter=0.3, training=True, grads=True),
    'mlp_w64_train_sharp': dict(kind='mlp', width=64, n_rays=16, jitter=0.1, beta=0.01, training=True, grads=True),
    'mlp_w64_image_eval': dict(kind='mlp', width=64, n_rays=20, jitter=0.3, training=False, image_mode=True),
    # image-mode (uv + pose + intrinsics) input AND training together: the yardstick case of the uv-input training rows
    # of tests/test_gpu_shapes.py (round 3 held them to hand-written bars)
    'mlp_w64_image_train': dict(kind='mlp', width=64, n_rays=20, jitter=0.3, training=True, grads=True, image_mode=True),
    'mlp_w64_white_train': dict(kind='mlp', width=64, n_rays=12, jitter=0.3, training=True, grads=True,
                                white_bkgd=True),
    'mlp_w64_code_train': dict(kind='mlp', width=64, n_rays=12, jitter=0.3, training=True, grads=True,
                               per_image_code=True),
    # if_hdr = True (6 of the reference's confs): ReLU instead of the sigmoid on the colour output
    'mlp_w64_hdr_train': dict(kind='mlp', width=64, n_rays=12, jitter=0.3, training=True, grads=True, if_hdr=True),
    'mlp_w64_hdr_eval': dict(kind='mlp', width=64, n_rays=12, jitter=0.3, training=False, if_hdr=True, ray_seed=4,
                             trace=True),
    'mlp_w64_nerf_train': dict(kind='mlp', width=64, n_rays=12, jitter=0.3, training=True, grads=True,
                               render_mode='nerf'),
    'gridless_w128_train': dict(kind='gridless', width=128, n_rays=8, jitter=0.3, training=True, grads='digest'),
    'grid_small_train': dict(kind='grid', width=64, n_rays=12, jitter=0.3, training=True, grads=True,
                             num_levels=4, logmap=10, end_size=64),
    'grid_small_eval': dict(kind='grid', width=64, n_rays=12, jitter=0.3, training=False,
                            num_levels=4, logmap=10, end_size=64),
    'mlp_w256_eval': dict(kind='mlp', width=256, n_rays=8, jitter=0.05, training=False),
    'mlp_w256_train': dict(kind='mlp', width=256, n_rays=8, jitter=0.05, training=True, grads='digest'),
    # 128 rays x 98 samples + eikonal points = 12,800 points: 200 workgroups of the MLP kernels, ragged last tile
    'mlp_w256_train_r128': dict(kind='mlp', width=256, n_rays=128, jitter=0.05, training=True, grads='digest',
                                ray_seed=9),
    # the reference's full grid configuration (16 levels, 2^19 entries per hashed level, scannetGrids.conf)
    'grid_full_eval': dict(kind='grid', width=256, n_rays=8, jitter=0.3, training=False, ray_seed=5),
    # ... and a TRAINING step of it (BASELINE.json configs[2] at 32 rays): every parameter gradient as a digest, the
    # 12.2 M-float table gradient included -- the reference's double-backward wiring (hashgrid.py:71-101) at hashed
    # levels and the full table size, over the restated kernels
    'grid_full_train': dict(kind='grid', width=256, n_rays=32, jitter=0.3, training=True, grads='digest', ray_seed=6),
}


def digest(t):
    """Per-tensor checksum used where the full gradient would make the file large."""
    a = t.detach().double().flatten()
    idx = torch.linspace(0, a.numel() - 1, min(16, a.numel())).long()
    return np.concatenate([[a.sum().item(), a.abs().sum().item(), (a * a).sum().item()], a[idx].numpy()])


def run_case(name, spec):
    conf = conf_from_spec(spec)
    state = synth.make_state(conf, seed=spec.get('weight_seed', 0), jitter=spec['jitter'],
                             sdf_scale=spec.get('sdf_scale', 1.0))
    inputs, indices = make_inputs(spec)
    pixel = not spec.get('image_mode', False)
    model = ref_loader.build_model(conf, state, training=spec['training'], if_hdr=spec.get('if_hdr', False))
    rounds = [0]
    orig = model.implicit_network.get_sdf_vals

    def counted(x):
        rounds[0] += 1
        return orig(x)
    model.implicit_network.get_sdf_vals = counted
    torch.manual_seed(spec.get('torch_seed', 1234))
    with ref_loader.record_rng() as log, ref_loader.record_sampler(model) as trace:
        out = model({k: v.clone() for k, v in inputs.items()}, indices, if_pixel_input=pixel)
    noise = ref_loader.noise_from_log(log)
    rec = {'spec': np.frombuffer(repr(sorted(spec.items())).encode(), dtype=np.uint8),
           'rounds': np.asarray(rounds[0]), 'indices': indices.numpy()}
    assert len(trace) == rounds[0]
    # did the loop end because every ray's beta reached beta0, or at max_total_iters (ray_sampler.py:125,179)?
    beta0 = float(model.density.get_beta())
    rec['converged'] = np.asarray(bool(trace[-1]['beta'].max() <= beta0))
    if spec.get('trace'):
        for r, t in enumerate(trace):
            for k, v in t.items():
                rec['smp.r%d.%s' % (r, k)] = v.numpy()
    for k, v in inputs.items():
        rec['in.' + k] = v.numpy()
    for k, v in noise.items():
        rec['noise.' + k] = v.numpy()
    for k, v in out.items():
        rec['out.' + k] = v.detach().numpy()
    if spec.get('grads'):
        loss = mo.probe_loss(out)
        names = [n for n, _ in model.named_parameters()]
        grads = torch.autograd.grad(loss, [p for _, p in model.named_parameters()], allow_unused=True)
        rec['loss'] = np.asarray(loss.item())
        for n, g in zip(names, grads):
            if g is None:
                continue
            rec[('gdig.' if spec['grads'] == 'digest' else 'grad.') + n] = \
                digest(g) if spec['grads'] == 'digest' else g.numpy()
            if spec['grads'] == 'digest' and g.numel() > 1000000 and n.endswith('encoding.embeddings'):
                # the 12.2 M-float table gradient: a fingerprint that sees where every contribution landed
                offsets = model.implicit_network.encoding.offsets
                for k, v in synth.table_fingerprint(g, offsets).items():
                    rec['gtab.%s.%s' % (n, k)] = v
    return rec


def run_stages():
    """Stage-level vectors straight from the reference's sub-modules."""
    net = ref_loader.load()
    import model.ray_sampler as rs          # reference modules (imported by ref_loader)
    import model.density as dn
    import model.embedder as emb
    rec = {}
    g = torch.Generator().manual_seed(7)
    conf = config.mlp_config(64, 8)
    state = synth.make_state(conf, seed=3, jitter=0.3)
    model = ref_loader.build_model(conf, state, training=False)
    rays = synth.make_rays(10, seed=5)
    o, d = rays['ray_cam_loc'], rays['ray_dirs']
    # uniform sampler, eval + train(jitter recorded)
    us = rs.UniformSampler(1.1, 0.0, 128, take_sphere_intersection=True)
    z, near, far = us.get_z_vals(d, o, model)
    rec.update({'uni.o': o.numpy(), 'uni.d': d.numpy(), 'uni.z_eval': z.numpy(), 'uni.far': far.numpy()})
    model.train(True)
    with ref_loader.record_rng() as log:
        zt, _, _ = us.get_z_vals(d, o, model)
    model.train(False)
    rec.update({'uni.jitter': log[0][1].numpy(), 'uni.z_train': zt.numpy()})
    # rays that miss / graze the cube
    o2 = torch.tensor([[3.0, 0.0, 0.0], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [2.0, 2.0, 0.0]])
    d2 = torch.nn.functional.normalize(torch.tensor([[0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.5, 0.5, 0.7], [-1.0, -1.0, 0.0]]), dim=1)
    n2, f2 = us.near_far_from_cube(o2, d2, 1.1)
    rec.update({'cube.o': o2.numpy(), 'cube.d': d2.numpy(), 'cube.near': n2.numpy(), 'cube.far': f2.numpy()})
    # density (scalar and per-ray beta)
    dens = dn.LaplaceDensity(params_init={'beta': 0.1}, beta_min=0.0001)
    s = torch.randn(6, 40, generator=g) * 0.3
    b = torch.rand(6, 1, generator=g) * 0.2 + 0.01
    rec.update({'dens.sdf': s.numpy(), 'dens.beta': b.numpy(), 'dens.scalar': dens(s).detach().numpy(),
                'dens.perray': dens(s, beta=b).detach().numpy(), 'dens.get_beta': dens.get_beta().detach().numpy()})
    # error bound
    sampler = model.ray_sampler
    zz = torch.sort(torch.rand(6, 40, generator=g) * 3.0, -1)[0]
    dist = zz[:, 1:] - zz[:, :-1]
    dstar = torch.rand(6, 39, generator=g) * 0.2
    eb = sampler.get_error_bound(b, model, s.reshape(-1, 1), zz, dist, dstar)
    rec.update({'eb.z': zz.numpy(), 'eb.dstar': dstar.numpy(), 'eb.out': eb.detach().numpy()})
    # volume rendering weights
    w = model.volume_rendering(zz, s.reshape(-1, 1))
    rec.update({'vr.weights': w.detach().numpy()})
    # embedder
    fn, ch = emb.get_embedder(6, input_dims=3)
    x = torch.randn(9, 3, generator=g)
    rec.update({'pe.x': x.numpy(), 'pe.out6': fn(x).numpy()})
    fn4, _ = emb.get_embedder(4)
    rec['pe.out4'] = fn4(x).numpy()
    # sdf network pieces
    pts = torch.randn(33, 3, generator=g) * 0.7
    sdf, feat, grad = model.implicit_network.get_outputs(pts.clone())
    rec.update({'net.pts': pts.numpy(), 'net.sdf': sdf.detach().numpy(), 'net.feat': feat.detach().numpy(),
                'net.grad': grad.detach().numpy(),
                'net.sdf_vals': model.implicit_network.get_sdf_vals(pts.clone()).detach().numpy(),
                'net.grad_unclamped': model.implicit_network.gradient_sdf(pts.clone()).detach().numpy()})
    vd = torch.nn.functional.normalize(torch.randn(33, 3, generator=g), dim=1)
    rgb = model.rendering_network(pts, grad.detach(), vd, feat.detach(), torch.arange(33))['rgb']
    rec.update({'col.dirs': vd.numpy(), 'col.rgb': rgb.detach().numpy()})
    return rec


PLUMBING = dict(kind='mlp', width=256, n_rays=512, n_samples=64, jitter=0.3, ray_seed=1, weight_seed=0)


def run_plumbing():
    """BASELINE.json configs[0] ("512 rays x 64 uniform samples, fp32 PyTorch CPU, plumbing"): the reference's own
    UniformSampler + ImplicitNetwork.get_outputs + RenderingNetwork + volume_rendering + composites, composed the
    way MonoSDFNetwork.forward composes them (network.py:532-562,603-611), at the 8x256 model of configs[1]."""
    ref_loader.load()
    import model.ray_sampler as rs          # reference module
    spec = PLUMBING
    conf = conf_from_spec(spec)
    state = synth.make_state(conf, seed=spec['weight_seed'], jitter=spec['jitter'])
    model = ref_loader.build_model(conf, state, training=False)
    rays = synth.make_rays(spec['n_rays'], seed=spec['ray_seed'], random_pose=True)
    dirs, cam = rays['ray_dirs'], rays['ray_cam_loc']
    us = rs.UniformSampler(conf['scene_bounding_sphere'], 0.0, spec['n_samples'], take_sphere_intersection=True)
    z, near, far = us.get_z_vals(dirs, cam, model)
    n, s = z.shape
    pts = (cam.unsqueeze(1) + z.unsqueeze(2) * dirs.unsqueeze(1)).reshape(-1, 3)
    dirs_flat = dirs.unsqueeze(1).repeat(1, s, 1).reshape(-1, 3)
    sdf, feat, grad = model.implicit_network.get_outputs(pts)
    rgb = model.rendering_network(pts, grad, dirs_flat, feat, torch.arange(n))['rgb'].reshape(-1, s, 3)
    w = model.volume_rendering(z, sdf)
    rgb_values = torch.sum(w.unsqueeze(-1) * rgb, 1)
    depth = torch.sum(w * z, 1, keepdims=True) / (w.sum(dim=1, keepdims=True) + 1e-8)
    depth = rays['ray_dirs_tmp'][:, 2:] * depth
    normals = grad / (grad.norm(2, -1, keepdim=True) + 1e-6)
    nmap = torch.sum(w.unsqueeze(-1) * normals.reshape(-1, s, 3), 1)
    rot = rays['ray_pose'][:, :3, :3].transpose(1, 2)
    nmap = (rot @ nmap.unsqueeze(-1)).squeeze(-1)
    k = 8                                     # per-sample tensors: every 8th ray
    d = lambda t: t.detach().numpy()
    return {'spec': np.frombuffer(repr(sorted(spec.items())).encode(), dtype=np.uint8),
            'out.z_vals': d(z), 'out.far': d(far), 'out.rgb_values': d(rgb_values), 'out.depth_values': d(depth),
            'out.normal_map': d(nmap), 'sub.sdf': d(sdf.reshape(n, s)[::k]), 'sub.weights': d(w[::k]),
            'sub.rgb': d(rgb[::k])}


def main(argv):
    os.makedirs(OUT_DIR, exist_ok=True)
    only = set(argv[1:])
    for name, spec in CASES.items():
        if only and name not in only:
            continue
        rec = run_case(name, spec)
        path = os.path.join(OUT_DIR, name + '.npz')
        np.savez_compressed(path, **rec)
        print('%-24s rounds=%d converged=%d  %6.1f KB' % (name, int(rec['rounds']), int(rec['converged']),
                                                         os.path.getsize(path) / 1024))
    if not only or 'plumbing' in only:
        path = os.path.join(OUT_DIR, 'plumbing_uniform64.npz')
        np.savez_compressed(path, **run_plumbing())
        print('%-24s %6.1f KB' % ('plumbing_uniform64', os.path.getsize(path) / 1024))
    if not only or 'stages' in only:
        path = os.path.join(OUT_DIR, 'stages.npz')
        np.savez_compressed(path, **run_stages())
        print('%-24s %6.1f KB' % ('stages', os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main(sys.argv)
