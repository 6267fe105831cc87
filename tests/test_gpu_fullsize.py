"""Full-size checks (BASELINE.json configs[1] / configs[2] shapes: 1024 rays x 98 samples, 8x256 SDF MLP; 16x2
hash grid at 104,448 points) through size-independent properties -- the oracle takes minutes at this size:
partition of unity of the compositing weights, composites recomputed from the per-sample outputs,
run-to-run bitwise reproducibility, a central finite difference of the loss against the analytic parameter
gradient, agreement of the two matrix cores, linearity / adjointness of the hash-grid kernels."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402  (model_conf / make_rays of the bench workload)

pytestmark = pytest.mark.gpu
N = bench.N_RAYS


def _model(precision, seed=0):
    from monosdf_amd.model.network import MonoSDFNetwork
    torch.manual_seed(seed)
    return MonoSDFNetwork(bench.model_conf()).cuda().train().set_precision(precision)


def _step(model, rays, seed=7):
    from monosdf_amd import ops
    torch.manual_seed(seed)                    # the same six random draws every call
    model.zero_grad(set_to_none=True)
    out = model(rays, torch.arange(N, device='cuda'), if_pixel_input=True)
    loss = ops.probe_loss(out)
    loss.backward()
    return out, loss


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3', 'bf16x6'])
def test_fullsize_compositing_identities_and_reproducibility(precision):
    model = _model(precision)
    rays = bench.make_rays(N, 1, 'cuda')
    out, loss = _step(model, rays)
    w, z = out['weights'], out['z_vals']
    assert w.shape == (N, 98) and out['rgb'].shape == (N, 98, 3)
    assert (w >= 0).all()
    assert (w.sum(1) - 1).abs().max().item() < 2e-5            # last interval is 1e10: the weights sum to one
    rgb_values = (w.unsqueeze(-1) * out['rgb']).sum(1)
    assert (rgb_values - out['rgb_values']).abs().max().item() < 2e-6
    depth = (w * z).sum(1, keepdim=True) / (w.sum(1, keepdim=True) + 1e-8) * rays['ray_dirs_tmp'][:, 2:]
    assert (depth - out['depth_values']).abs().max().item() < 2e-5
    assert (out['depth_vals'] - z * rays['ray_dirs_tmp'][:, 2:]).abs().max().item() == 0.0
    g1 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    out2, loss2 = _step(model, rays)
    assert loss.item() == loss2.item()
    for k in ('rgb_values', 'depth_values', 'normal_map', 'sdf', 'weights', 'grad_theta'):
        assert torch.equal(out[k], out2[k]), k                 # no float atomics on this path: bitwise reproducible
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, g1[n]), n


def _central_difference(p, d, eps, value_fn):
    vals = []
    for s in (+1.0, -1.0):
        with torch.no_grad():
            p.add_(s * eps * d)
            vals.append(value_fn())
            p.sub_(s * eps * d)
    return (vals[0] - vals[1]) / (2 * eps)


def test_fullsize_directional_derivative():
    """(L(theta + e d) - L(theta - e d)) / 2e  vs  <grad L, d>.
    Colour-network parameters through the whole step (they do not move the samples); SDF-network parameters
    through get_outputs on fixed points (first- and second-order paths) -- in the full step the sampler places
    the samples from the SDF under no_grad, as the reference does, so a finite difference would see that too."""
    from monosdf_amd import ops
    model = _model('fp32')
    rays = bench.make_rays(N, 1, 'cuda')
    _step(model, rays)
    params = dict(model.named_parameters())
    idx = torch.arange(N, device='cuda')

    def step_loss():
        torch.manual_seed(7)
        return ops.probe_loss(model(rays, idx, if_pixel_input=True)).item()

    gen = torch.Generator(device='cuda').manual_seed(3)
    for name in ('rendering_network.lin1.bias', 'rendering_network.lin0.weight_g'):
        p = params[name]
        # direction = the gradient's own plus half a random one: the loss (~0.5) is an fp32 scalar, so a central
        # difference over 2 eps resolves derivatives down to ~6e-8 / 4e-3 = 1.5e-5 -- a purely random direction in
        # a 256-dimensional parameter has a derivative of that size (the test then compares rounding noise)
        r = torch.randn(p.shape, device='cuda', generator=gen)
        d = p.grad / p.grad.norm() + 0.5 * r / r.norm()
        d /= d.norm()
        analytic = (p.grad * d).sum().item()
        numeric = _central_difference(p, d, 2e-3, step_loss)
        assert abs(numeric - analytic) <= 0.02 * abs(analytic) + 2e-6, (name, numeric, analytic)

    P = N * 102
    x = (torch.rand(P, 3, device='cuda', generator=gen) * 2 - 1) * 0.9
    ca = torch.randn(P, 1, device='cuda', generator=gen) / P
    cb = torch.randn(P, 256, device='cuda', generator=gen) * 0.1 / P
    cc = torch.randn(P, 3, device='cuda', generator=gen) / P

    def probe(with_grad):
        # the gradient term uses gradient_sdf (unclamped sdf, network.py:98-109): get_outputs' min(sdf, sphere)
        # makes d sdf/dx JUMP when a point changes sides, which a few of the 104,448 points do under any
        # finite step (measured: each such point moves the difference quotient by O(1)/(P eps))
        sdf, feat, _ = model.implicit_network.get_outputs(x)
        grad = model.implicit_network.gradient_sdf(x)
        # fp64 sums: the fp32 reduction noise of 27M terms would be as large as the finite difference
        return ((ca.double() * sdf.double()).sum() + (cb.double() * feat.double()).sum() +
                (cc.double() * grad.double()).sum())

    model.zero_grad(set_to_none=True)
    probe(True).backward()
    for name in ('implicit_network.lin3.bias', 'implicit_network.lin6.weight_g', 'implicit_network.lin0.weight_v'):
        p = params[name]
        d = p.grad / p.grad.norm()                      # steepest direction: the largest signal per unit step
        analytic = (p.grad * d).sum().item()
        # 1e-4: Softplus(beta=100) bends over ~1e-2 of pre-activation; steps of 1e-3 / 3e-4 along the steepest
        # direction already show its curvature (70 % / 4 % off), the fp64 probe keeps 1e-4 above the noise
        numeric = _central_difference(p, d, 1e-4, lambda: probe(False).item())
        assert abs(numeric - analytic) <= 0.02 * abs(analytic) + 1e-7, (name, numeric, analytic)


def test_fullsize_matrix_cores_agree():
    rays = bench.make_rays(N, 1, 'cuda')
    res = {}
    for precision in ('fp32', 'bf16x3'):
        model = _model(precision)
        out, loss = _step(model, rays)
        res[precision] = (out, loss.item(), {n: p.grad.clone() for n, p in model.named_parameters()
                                             if p.grad is not None})
    (o32, l32, g32), (o16, l16, g16) = res['fp32'], res['bf16x3']
    assert abs(l32 - l16) <= 1e-4 * max(1.0, abs(l32))
    rel = lambda a, b: ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()
    for k in ('rgb_values', 'depth_values', 'normal_map', 'grad_theta'):
        assert rel(o16[k], o32[k]) < 5e-4, (k, rel(o16[k], o32[k]))      # sampler tolerance of test_gpu_parity
    for n in g32:
        assert rel(g16[n], g32[n]) < 2e-3, (n, rel(g16[n], g32[n]))


def test_fullsize_hash_grid_linearity_and_adjointness():
    """forward is linear in the table; backward (LDS path for the two coarsest levels + atomics) is its adjoint."""
    from monosdf_amd.hashencoder.hashgrid import HashEncoder
    enc = HashEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                      desired_resolution=2048).cuda()
    B = N * 102
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.rand(B, 3, device='cuda', generator=g) * 2 - 1
    with torch.no_grad():
        enc.embeddings.copy_(torch.randn(enc.embeddings.shape, device='cuda', generator=g) * 0.1)
    y = enc(x)
    g1 = torch.randn(y.shape, device='cuda', generator=g)
    g2 = torch.randn(y.shape, device='cuda', generator=g)
    ga, = torch.autograd.grad(y, enc.embeddings, g1, retain_graph=True)
    gb, = torch.autograd.grad(y, enc.embeddings, g2, retain_graph=True)
    gab, = torch.autograd.grad(y, enc.embeddings, g1 + g2)
    scale = gab.abs().max().item()
    assert (ga + gb - gab).abs().max().item() < 1e-4 * scale              # linearity (atomics: order noise only)
    lhs = (y.double() * g1.double()).sum().item()                        # <E(theta), g>
    rhs = (enc.embeddings.double() * ga.double()).sum().item()          # <theta, E^T g>
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0), (lhs, rhs)
    # coarsest levels: every one of their entries receives gradient from ~100k points
    n0 = int(enc.offsets[2].item())
    assert (ga[:n0].abs().sum(1) > 0).float().mean().item() > 0.5


def test_fullsize_binned_scatter_equals_atomic_scatter():
    """configs[2] size (104,448 points, 16 x 2, 2^19 entries per hashed level): the binned scatter (crowded bins are
    cut over several workgroups, the coarse levels hold hundreds of contributions per entry) against the
    one-atomic-per-corner kernels, first order, second order and both fused."""
    from monosdf_amd import _lib
    from monosdf_amd.hashencoder.hashgrid import HashEncoder
    enc = HashEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                      desired_resolution=2048).cuda()
    B, L, C = N * 102, 16, 2
    g = torch.Generator(device='cuda').manual_seed(6)
    # half of the points on 64 "rays" through a small region (crowded cells on every level), half anywhere
    t = torch.rand(B // 2, 1, device='cuda', generator=g)
    o = torch.rand(64, 3, device='cuda', generator=g)[torch.randint(64, (B // 2,), device='cuda', generator=g)]
    x = torch.cat([0.45 + 0.1 * (o + t * 0.3), torch.rand(B - B // 2, 3, device='cuda', generator=g)]).contiguous()
    grad = torch.randn(L, B, C, device='cuda', generator=g)
    grad2 = torch.randn(L, B, C, device='cuda', generator=g)
    gg = torch.randn(B, 3, device='cuda', generator=g)
    emb, offs = enc.embeddings.detach(), enc.offsets
    n = emb.shape[0]
    S, H = enc.log2_scale, int(enc.base_resolution)
    st = _lib.stream_ptr()
    dy = torch.empty(B, L * 3 * C, device='cuda')
    out = torch.empty(L, B, C, device='cuda')
    _lib.call('msdf_hash_encode_forward', _lib.ptr(x), _lib.ptr(emb), _lib.ptr(offs), _lib.ptr(out), B, 3, C, L, S, H, 1,
              _lib.ptr(dy), st)
    a1, a2 = torch.zeros_like(emb), torch.zeros_like(emb)
    gi, ggrad = torch.zeros_like(x), torch.zeros(L, B, C, device='cuda')
    _lib.call('msdf_hash_encode_backward', _lib.ptr(grad), _lib.ptr(x), _lib.ptr(emb), _lib.ptr(offs), _lib.ptr(a1),
              B, 3, C, L, S, H, 0, _lib.ptr(dy), _lib.ptr(gi), st)
    _lib.call('msdf_hash_encode_second_backward', _lib.ptr(grad2), _lib.ptr(x), _lib.ptr(emb), _lib.ptr(offs), B, 3, C, L,
              S, H, 1, _lib.ptr(dy), _lib.ptr(gg), _lib.ptr(ggrad), _lib.ptr(a2), st)
    nbytes = _lib.load().msdf_hash_scatter_workspace_bytes(B, C, L, n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    b1, b2, bf = torch.zeros_like(emb), torch.zeros_like(emb), torch.zeros_like(emb)
    _lib.call('msdf_hash_encode_backward_ws', _lib.ptr(grad), _lib.ptr(x), _lib.ptr(emb), _lib.ptr(offs), _lib.ptr(b1),
              B, 3, C, L, S, H, 0, _lib.ptr(dy), None, n, _lib.ptr(ws), nbytes, st)
    _lib.call('msdf_hash_encode_second_backward_ws', _lib.ptr(grad2), _lib.ptr(x), _lib.ptr(emb), _lib.ptr(offs), B, 3, C,
              L, S, H, 1, _lib.ptr(dy), _lib.ptr(gg), None, _lib.ptr(b2), n, _lib.ptr(ws), nbytes, st)
    _lib.call('msdf_hash_encode_backward_fused', _lib.ptr(grad), _lib.ptr(grad2), _lib.ptr(x), _lib.ptr(offs), _lib.ptr(bf),
              B, 3, C, L, S, H, _lib.ptr(gg), n, _lib.ptr(ws), nbytes, st)
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    assert a1.abs().max().item() > 0 and a2.abs().max().item() > 0
    # both sides sum thousands of signed fp32 terms per crowded entry in an arbitrary order: 1e-4 of the largest entry
    assert rel(b1, a1) < 1e-4, rel(b1, a1)
    assert rel(b2, a2) < 1e-4, rel(b2, a2)
    assert rel(bf, a1 + a2) < 1e-4, rel(bf, a1 + a2)
    # per level: same support (an entry is touched by the binned path exactly when the atomics touch it)
    assert torch.equal(b1 != 0, a1 != 0)


def test_fullsize_scatter_of_ray_ordered_samples_and_output_form():
    """The place kernel of the binned scatter sums consecutive points that sit in one cell before it writes a record
    (runs inside 16-lane rows).  Ray-ordered samples 2e-4 apart give runs of every length on every level, some points
    outside [0, 1] (no contribution) break them; against the one-atomic-per-corner kernels.  And the "=" form
    (msdf_hash_encode_backward_fused_out) into an uninitialised buffer equals the "+=" form into zeros."""
    from monosdf_amd import _lib
    from monosdf_amd.hashencoder.hashgrid import HashEncoder
    enc = HashEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                      desired_resolution=2048).cuda()
    n_rays, per = 400, 98
    B, L, C = n_rays * per + 37, 16, 2
    g = torch.Generator(device='cuda').manual_seed(9)
    o = torch.rand(n_rays, 1, 3, device='cuda', generator=g) * 0.8 + 0.1
    d = torch.nn.functional.normalize(torch.randn(n_rays, 1, 3, device='cuda', generator=g), dim=-1)
    # step sizes from 2e-5 (runs of 64+ at the coarse levels) to 5e-3 per ray
    step = (2e-5 * (250.0 ** torch.rand(n_rays, 1, 1, device='cuda', generator=g)))
    t = torch.arange(per, device='cuda').view(1, per, 1) * step
    x = (o + t * d).reshape(-1, 3)
    x = torch.cat([x, torch.rand(37, 3, device='cuda', generator=g)])
    x[5::97] = 1.5                                  # out of range: contributes nothing, breaks a run
    x[1000:1100] = x[1000]                          # 100 identical points across rows and waves
    x = x.contiguous()
    grad = torch.randn(L, B, C, device='cuda', generator=g)
    grad2 = torch.randn(L, B, C, device='cuda', generator=g)
    gg = torch.randn(B, 3, device='cuda', generator=g)
    emb, offs = enc.embeddings.detach(), enc.offsets
    n = emb.shape[0]
    S, H = enc.log2_scale, int(enc.base_resolution)
    st = _lib.stream_ptr()
    dy = torch.empty(B, L * 3 * C, device='cuda')
    out = torch.empty(L, B, C, device='cuda')
    _lib.call('msdf_hash_encode_forward', _lib.ptr(x), _lib.ptr(emb), _lib.ptr(offs), _lib.ptr(out), B, 3, C, L, S, H, 1,
              _lib.ptr(dy), st)
    a1, a2 = torch.zeros_like(emb), torch.zeros_like(emb)
    gi, ggrad = torch.zeros_like(x), torch.zeros(L, B, C, device='cuda')
    _lib.call('msdf_hash_encode_backward', _lib.ptr(grad), _lib.ptr(x), _lib.ptr(emb), _lib.ptr(offs), _lib.ptr(a1),
              B, 3, C, L, S, H, 0, _lib.ptr(dy), _lib.ptr(gi), st)
    _lib.call('msdf_hash_encode_second_backward', _lib.ptr(grad2), _lib.ptr(x), _lib.ptr(emb), _lib.ptr(offs), B, 3, C, L,
              S, H, 1, _lib.ptr(dy), _lib.ptr(gg), _lib.ptr(ggrad), _lib.ptr(a2), st)
    nbytes = _lib.load().msdf_hash_scatter_workspace_bytes(B, C, L, n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    b1, bf = torch.zeros_like(emb), torch.zeros_like(emb)
    bo = torch.full_like(emb, float('nan'))
    _lib.call('msdf_hash_encode_backward_ws', _lib.ptr(grad), _lib.ptr(x), _lib.ptr(emb), _lib.ptr(offs), _lib.ptr(b1),
              B, 3, C, L, S, H, 0, _lib.ptr(dy), None, n, _lib.ptr(ws), nbytes, st)
    _lib.call('msdf_hash_encode_backward_fused', _lib.ptr(grad), _lib.ptr(grad2), _lib.ptr(x), _lib.ptr(offs), _lib.ptr(bf),
              B, 3, C, L, S, H, _lib.ptr(gg), n, _lib.ptr(ws), nbytes, st)
    _lib.call('msdf_hash_encode_backward_fused_out', _lib.ptr(grad), _lib.ptr(grad2), _lib.ptr(x), _lib.ptr(offs),
              _lib.ptr(bo), B, 3, C, L, S, H, _lib.ptr(gg), n, _lib.ptr(ws), nbytes, st)
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    assert rel(b1, a1) < 1e-4, rel(b1, a1)
    assert rel(bf, a1 + a2) < 1e-4, rel(bf, a1 + a2)
    assert torch.equal(b1 != 0, a1 != 0)
    assert torch.isfinite(bo).all()
    assert rel(bo, a1 + a2) < 1e-4, rel(bo, a1 + a2)
    # per level, so that a fine level's error cannot hide under a coarse level's magnitude
    for l in range(L):
        lo, hi = int(offs[l]), int(offs[l + 1])
        assert rel(bo[lo:hi], (a1 + a2)[lo:hi]) < 2e-4, (l, rel(bo[lo:hi], (a1 + a2)[lo:hi]))


def _record(name, payload):
    """Times of the full-size runs go to gpurun_out/r02_fullsize.json (copied to profiles/ by hand)."""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'r02_fullsize.json')
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[name] = payload
        json.dump(data, open(path, 'w'), indent=1)
    except OSError:
        pass


def test_fullsize_image_render_384():
    """BASELINE.json configs[3] at full size on one GPU: 384 x 384 pixels in 144 chunks of 1,024 rays (image mode:
    uv + pose + intrinsics), 8x256 network.  Property: the chunked image equals the same pixels rendered in chunks of
    another size (rays are independent; at the random-init state every chunk's sampler converges in one round)."""
    import time
    from monosdf_amd.utils import render
    model = _model('fp32').eval()
    side = 384
    ys, xs = torch.meshgrid(torch.arange(side), torch.arange(side), indexing='ij')
    uv = torch.stack([xs.flatten(), ys.flatten()], -1)[None].float().cuda() + 0.5
    intr = torch.eye(4)[None].clone()
    intr[0, 0, 0] = intr[0, 1, 1] = 300.0
    intr[0, 0, 2] = intr[0, 1, 2] = side / 2
    pose = torch.eye(4)[None].clone()
    pose[0, :3, 3] = torch.tensor([0.05, -0.1, 0.15])
    inputs = {'uv': uv, 'pose': pose.cuda(), 'intrinsics': intr.cuda()}
    idx = torch.zeros(1, dtype=torch.long, device='cuda')
    total = side * side
    render.render_image(model, {**inputs, 'uv': uv[:, :2048]}, idx, 2048, split_n_pixels=1024)     # warm-up
    torch.cuda.synchronize()
    t0 = time.time()
    img = render.render_image(model, inputs, idx, total, split_n_pixels=1024)
    torch.cuda.synchronize()
    dt = time.time() - t0
    assert img['rgb_values'].shape == (total, 3) and img['depth_values'].shape == (total, 1)
    assert torch.isfinite(img['rgb_values']).all() and (img['rgb_values'] >= 0).all() and (img['rgb_values'] <= 1).all()
    # a pixel subsample rendered on its own, in chunks of 500: same values
    g = torch.Generator().manual_seed(0)
    pick = torch.randperm(total, generator=g)[:3000].cuda()
    sub = render.render_image(model, {**inputs, 'uv': uv[:, pick]}, idx, pick.numel(), split_n_pixels=500)
    for k in sub:
        assert (sub[k] - img[k][pick]).abs().max().item() <= 1e-5 * max(1.0, img[k].abs().max().item()), k
    _record('configs[3] 384x384 render, 144 chunks of 1024 rays, one GPU, eval mode',
            {'seconds': dt, 'rays_per_second': total / dt, 'sampler_rounds_last_chunk': model.ray_sampler.last_rounds})


def test_fullsize_sdf_volume_512():
    """BASELINE.json configs[4] at full size on one GPU: the 512^3 coarse-to-fine SDF volume (one block, pyramid
    64^3 -> 512^3).  Properties: every voxel the pyramid refined to the finest level holds the network's value at its
    own centre; every other voxel holds the value of the coarser voxel that covers it."""
    import time
    import numpy as np
    from monosdf_amd.utils import render
    model = _model('fp32').eval()
    fn = lambda p: model.implicit_network(p)[:, 0]
    with torch.no_grad():
        fn(torch.zeros(64, 3, device='cuda'))
        torch.cuda.synchronize()
        t0 = time.time()
        blocks = list(render.sdf_volume(fn, resolution=512, grid_boundary=(-1.1, 1.1), shard=False))
        torch.cuda.synchronize()
        dt_first = time.time() - t0           # includes the one-off allocation of the 537 MB pinned buffer
        t0 = time.time()
        blocks = list(render.sdf_volume(fn, resolution=512, grid_boundary=(-1.1, 1.1), shard=False))
        torch.cuda.synchronize()
        dt = time.time() - t0                 # what every further block costs (resolution 1024 = 8 such blocks)
        # the same volume from implicit_network.raw_sdf (column 0 alone: no features, no gradient)
        raw = lambda p: model.implicit_network.raw_sdf(p)
        t0 = time.time()
        (_, _, vol_raw), = list(render.sdf_volume(raw, resolution=512, grid_boundary=(-1.1, 1.1), shard=False))
        torch.cuda.synchronize()
        dt_raw = time.time() - t0
    # column 0 of the fused forward and the sdf-only kernel: the same network values (a voxel next to a refinement
    # threshold may still be refined by one and not the other: those are left out)
    diff = np.abs(vol_raw - blocks[0][2])
    assert float((diff > 1e-5).mean()) < 1e-3, float((diff > 1e-5).mean())          # |sdf| <= 2: a few ulp, plus threshold voxels
    assert len(blocks) == 1
    origin, spacing, vol = blocks[0]
    assert vol.shape == (512, 512, 512) and np.isfinite(vol).all()
    n = 512
    thr_fine = 2 * 2.2 / n * 8 / 8                       # threshold of the last refinement (plots.py:163,190)
    near = np.abs(vol) < thr_fine / 2                    # well inside it: certainly refined at every level
    frac = float(near.mean())
    assert 0 < frac < 0.2
    ii = np.argwhere(near)
    ii = ii[np.random.default_rng(0).permutation(len(ii))[:100000]]
    axis = np.linspace(-1.1, 1.1, n)
    pts = torch.from_numpy(np.stack([axis[ii[:, 0]], axis[ii[:, 1]], axis[ii[:, 2]]], 1)).float().cuda()
    with torch.no_grad():
        dense = fn(pts).cpu().numpy()
    got = vol[ii[:, 0], ii[:, 1], ii[:, 2]]
    assert np.abs(got - dense).max() <= 1e-5, np.abs(got - dense).max()
    # far from the surface the coarse value is replicated: 8x8x8 blocks of one value
    far_block = vol[:8, :8, :8]
    assert np.all(far_block == far_block[0, 0, 0])
    _record('configs[4] 512^3 SDF volume, coarse-to-fine, one GPU',
            {'seconds': dt, 'seconds_first_call': dt_first, 'seconds_with_raw_sdf': dt_raw, 'voxels': n ** 3,
             'fraction_refined_to_finest_level': frac})


def test_first_form_of_the_binned_scatter_still_runs():
    """The count / scan / place / accumulate form stays in the library as the path for tables of more than 8,192 slices
    per level; MSDF_HASH_SCATTER=1 (read once per process) selects it: the ray-ordered scatter test in a child process."""
    import subprocess
    import sys
    env = dict(os.environ, MSDF_HASH_SCATTER='1')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', os.path.abspath(__file__), '-k',
                        'ray_ordered or binned_scatter_equals'], cwd=root, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert '2 passed' in r.stdout, r.stdout[-500:]
