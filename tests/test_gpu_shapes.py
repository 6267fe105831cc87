"""Network geometries the goldens do not reach: widths that are not multiples of 16 (padded tiles), odd tile counts
(a product's last weight chunk holds ONE out tile), networks with and without the skip layer, 2 to 8 layers.
The fused kernels (forward + gradient, double backward with all weight gradients, full training forward) against the
CPU oracle -- which test_oracle_golden.py pins to the reference on the golden geometries; these geometries are the
same code of the oracle with other sizes."""
import pytest
import torch

from helpers import TOL, check, rel_err
from oracle import config, synth

pytestmark = pytest.mark.gpu

# (width, depth): 48 = 3 tiles (odd), 100 = 7 tiles with 12 padded slots, 176 = 11 tiles, 256 with 2 layers (the
# hash-grid configuration's MLP without the grid)
SHAPES = [(48, 2), (48, 5), (100, 3), (100, 8), (176, 5), (256, 2)]


def _build(width, depth, precision):
    from monosdf_amd.conf import ConfigTree
    from monosdf_amd.model.network import MonoSDFNetwork
    conf = config.mlp_config(width, depth)
    state = synth.make_state(conf, seed=width + depth, jitter=0.3)
    m = MonoSDFNetwork(ConfigTree.from_dict(conf))
    m.load_state_dict({k: v.clone() for k, v in state.items()}, strict=True)
    return conf, state, m.cuda().set_precision(precision)


# the bf16x6 core on the generic widths too (run-time K-block counts, chunks of one out tile): same bars, same rows
@pytest.mark.parametrize('precision', ['fp32', 'bf16x6'])
@pytest.mark.parametrize('width,depth', SHAPES)
def test_sdf_network_forward_gradient_and_double_backward(width, depth, precision, errlog):
    from oracle import monosdf_oracle as mo
    conf, state, m = _build(width, depth, precision)
    m.train()
    g = torch.Generator().manual_seed(11)
    P = 64 * 3 + 21                                    # ragged last tile
    x = (torch.rand(P, 3, generator=g) * 2 - 1) * 1.2
    F = conf['feature_vector_size']
    ca, cb, cc = torch.randn(P, 1, generator=g), torch.randn(P, F, generator=g) * 0.1, torch.randn(P, 3, generator=g)
    st = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in state.items()}
    sdf_o, feat_o, grad_o = mo.get_outputs(st, conf, x)
    sdf, feat, grad = m.implicit_network.get_outputs(x.cuda())
    assert rel_err(sdf, sdf_o) < TOL and rel_err(feat, feat_o) < TOL and rel_err(grad, grad_o) < TOL
    loss_o = (ca * sdf_o).sum() + (cb * feat_o).sum() + (cc * grad_o).sum()
    names = [n for n in st if n.startswith('implicit_network.lin')]
    g_o = torch.autograd.grad(loss_o, [st[n] for n in names])
    loss = (ca.cuda() * sdf).sum() + (cb.cuda() * feat).sum() + (cc.cuda() * grad).sum()
    loss.backward()
    params = dict(m.named_parameters())
    for n, go in zip(names, g_o):
        check(errlog, 'shapes_double_backward' + ('' if precision == 'fp32' else '.' + precision), '%dx%d' % (depth, width),
              n, rel_err(params[n].grad, go))


@pytest.mark.parametrize('width,depth', [(48, 5), (100, 3), (176, 5)])
def test_training_forward_and_gradients(width, depth, errlog):
    """The whole pass (sampler, both networks, compositor, eikonal block) and the gradients of the probe loss."""
    from oracle import monosdf_oracle as mo
    conf, state, m = _build(width, depth, 'fp32')
    m.train()
    n = 24
    rays = synth.make_rays(n, seed=3, random_pose=True)
    noise = synth.make_noise(conf, n, 128, seed=5)
    idx = torch.arange(n) % 7
    st = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in state.items()}
    ref = mo.render(st, conf, rays, idx, True, True, noise)
    m._noise = {k: v.cuda() for k, v in noise.items()}
    out = m({k: v.cuda() for k, v in rays.items()}, idx.cuda(), if_pixel_input=True)
    case = '%dx%d' % (depth, width)
    for k in ('rgb_values', 'depth_values', 'normal_map', 'weights', 'sdf', 'grad_theta'):
        check(errlog, 'shapes_training', case, k, rel_err(out[k], ref[k]))
    loss_o = mo.probe_loss(ref)
    names = [k for k, v in st.items() if v.requires_grad]
    g_o = dict(zip(names, torch.autograd.grad(loss_o, [st[k] for k in names], allow_unused=True)))
    mo.probe_loss(out).backward()
    for k, p in m.named_parameters():
        if g_o.get(k) is None:
            continue
        check(errlog, 'shapes_training', case, k, rel_err(p.grad, g_o[k]))


def test_too_wide_network_is_refused():
    """257 outputs (256 features + sdf) are 17 tiles, the register layout's maximum: 272 features do not fit."""
    with pytest.raises(RuntimeError, match='tile'):
        _, _, m = _build(272, 2, 'fp32')
        m.implicit_network.get_outputs(torch.zeros(64, 3, device='cuda'))


def _variant(edit, seed=21):
    from monosdf_amd.conf import ConfigTree
    from monosdf_amd.model.network import MonoSDFNetwork
    conf = config.mlp_config(64, 8)
    edit(conf)
    state = synth.make_state(conf, seed=seed, jitter=0.3)
    m = MonoSDFNetwork(ConfigTree.from_dict(conf))
    m.load_state_dict({k: v.clone() for k, v in state.items()}, strict=True)
    return conf, state, m.cuda()


def _set(section, **kw):
    def edit(conf):
        conf[section].update(kw)
    return edit


VARIANTS = {
    'no_positional_encoding': _set('implicit_network', multires=0),
    'multires_4': _set('implicit_network', multires=4),
    'skip_at_2': _set('implicit_network', skip_in=[2]),
    'no_skip': _set('implicit_network', skip_in=[]),
    'outside_in': _set('implicit_network', inside_outside=False),
    'view_pe_0': _set('rendering_network', multires_view=0),
    'view_pe_2': _set('rendering_network', multires_view=2),
    'colour_3_hidden': _set('rendering_network', dims=[64, 64, 64]),
    'colour_1_hidden': _set('rendering_network', dims=[64]),
}


def _relu_units_near_zero(fn, thresh=1e-6):
    """Runs fn() with torch.relu wrapped; returns (result, per relu call the boolean mask of the units whose input comes
    within `thresh` of zero for some sample).  Such a unit's ReLU switch can differ between two fp32 summation orders of
    the same dot product, and with it that sample's whole contribution to the unit's row of the weight gradient."""
    masks = []
    real = torch.relu

    def spy(h):
        masks.append((h.detach().abs().reshape(-1, h.shape[-1]).min(0)[0] < thresh))
        return real(h)
    torch.relu = spy
    try:
        res = fn()
    finally:
        torch.relu = real
    return res, masks


@pytest.mark.parametrize('name', sorted(VARIANTS))
def test_config_options_against_oracle(name, errlog):
    """Options of the reference's conf files the goldens leave at their usual values: one training pass (forward
    outputs and the gradients of the probe loss) against the oracle.  Bars: 1e-4 unless tests/golden/tolerances.json
    lists the comparison (`options|<variant>|<tensor>`)."""
    from oracle import monosdf_oracle as mo
    conf, state, m = _variant(VARIANTS[name], seed=21)
    m.train()
    n = 16
    rays = synth.make_rays(n, seed=6, random_pose=True)
    noise = synth.make_noise(conf, n, 128, seed=8)
    idx = torch.arange(n) % 5
    st = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in state.items()}
    # weight seed 21 puts one ReLU input of the colour network's first layer at 9.9e-8 in the 'no_skip' network: the rows
    # of such units are left out of that layer's gradient comparison (the unit is found in the ORACLE's pre-activations)
    ref, near_zero = _relu_units_near_zero(lambda: mo.render(st, conf, rays, idx, True, True, noise))
    m._noise = {k: v.cuda() for k, v in noise.items()}
    out = m({k: v.cuda() for k, v in rays.items()}, idx.cuda(), if_pixel_input=True)
    for k in ('rgb_values', 'depth_values', 'normal_map', 'weights', 'sdf', 'grad_theta'):
        check(errlog, 'options', name, k, rel_err(out[k], ref[k]))
    names = [k for k, v in st.items() if v.requires_grad]
    g_o = dict(zip(names, torch.autograd.grad(mo.probe_loss(ref), [st[k] for k in names], allow_unused=True)))
    mo.probe_loss(out).backward()
    for k, p in m.named_parameters():
        if g_o.get(k) is None:
            continue
        g, gr = p.grad.detach().cpu(), g_o[k]
        if k.startswith('rendering_network.lin'):
            layer = int(k.split('.')[1][3:])
            if layer < len(near_zero) and bool(near_zero[layer].any()):
                keep = ~near_zero[layer]
                assert int(keep.sum()) >= keep.numel() - 4, 'more than four units at a ReLU switch: pick another seed'
                g, gr = g[keep], gr[keep]
        check(errlog, 'options', name, k, rel_err(g, gr))


def test_wide_positional_encoding_is_refused():
    """multires = 10 is 63 encoding slots; the kernels hold 48 (multires <= 7; every conf of the reference uses 6)."""
    with pytest.raises(RuntimeError, match='positional encoding'):
        _, _, m = _variant(_set('implicit_network', multires=10))
        m.implicit_network.get_outputs(torch.zeros(64, 3, device='cuda'))


SAMPLER_VARIANTS = {
    'fewer_samples': dict(N_samples=32, N_samples_eval=64, N_samples_extra=16),
    'tighter_bound': dict(eps=0.05, beta_iters=6),
    'three_rounds_at_most': dict(max_total_iters=3),
    'near_offset': dict(near=0.1),
}


@pytest.mark.parametrize('name', sorted(SAMPLER_VARIANTS))
def test_sampler_options_against_oracle(name, errlog):
    """Sampler settings other than the ones every conf of the reference uses (64 / 128 / 32 samples, eps 0.1, 10
    bisection steps, 5 rounds): a sharp state (beta 0.01, 2+ rounds) in eval mode -- z_vals and the rendered values
    against the oracle -- or a clear refusal."""
    from oracle import monosdf_oracle as mo
    from monosdf_amd.conf import ConfigTree
    from monosdf_amd.model.network import MonoSDFNetwork
    conf = config.mlp_config(64, 8, beta=0.01)
    conf['ray_sampler'].update(SAMPLER_VARIANTS[name])
    state = synth.make_state(conf, seed=31, jitter=0.1)
    n = 12
    rays = synth.make_rays(n, seed=9, random_pose=True)
    idx = torch.arange(n) % 5
    try:
        m = MonoSDFNetwork(ConfigTree.from_dict(conf))
        m.load_state_dict({k: v.clone() for k, v in state.items()}, strict=True)
        m = m.cuda().eval()
        with torch.no_grad():
            out = m({k: v.cuda() for k, v in rays.items()}, idx.cuda(), if_pixel_input=True)
    except (RuntimeError, NotImplementedError, ValueError) as e:
        assert 'monosdf_amd' in str(e), e          # refused by this package, with a reason
        pytest.skip('refused: %s' % e)
    ref = mo.render({k: v.clone() for k, v in state.items()}, conf, rays, idx, True, False, None)   # needs autograd
    ref = {k: v.detach() for k, v in ref.items()}
    assert out['z_vals'].shape == ref['z_vals'].shape
    far = mo.sampler_far(conf)
    check(errlog, 'sampler_options', name, 'z_vals', (out['z_vals'].cpu() - ref['z_vals']).abs().max().item() / far)
    for k in ('rgb_values', 'depth_values'):
        check(errlog, 'sampler_options', name, k, rel_err(out[k], ref[k]))


GRID_VARIANTS = {
    '8_levels_x_4_features': dict(num_levels=8, level_dim=4, logmap=11, base_size=8, end_size=128),
    '3_levels_x_8_features': dict(num_levels=3, level_dim=8, logmap=10, base_size=8, end_size=48),
    '12_levels_x_2_features': dict(num_levels=12, level_dim=2, logmap=12, base_size=16, end_size=512),
}


@pytest.mark.parametrize('name', sorted(GRID_VARIANTS))
def test_hash_grid_model_variants_against_oracle(name, errlog):
    """Hash-grid models with other level counts / features per level than the reference's 16 x 2 (through the fused
    encoding + MLP node, embedding gradients included) against the oracle.  Hash arithmetic: parity unpinned by
    reference outputs (README) -- this pins the wiring for other table geometries."""
    from oracle import monosdf_oracle as mo
    from monosdf_amd.conf import ConfigTree
    from monosdf_amd.model.network import MonoSDFNetwork
    g = GRID_VARIANTS[name]
    conf = config.grid_config(64, 0.1, g['num_levels'], g['level_dim'], g['logmap'], g['base_size'], g['end_size'])
    state = synth.make_state(conf, seed=41, jitter=0.3)
    m = MonoSDFNetwork(ConfigTree.from_dict(conf))
    m.load_state_dict({k: v.clone() for k, v in state.items()}, strict=True)
    m = m.cuda().train()
    n = 12
    rays = synth.make_rays(n, seed=4, random_pose=True)
    noise = synth.make_noise(conf, n, 128, seed=6)
    idx = torch.arange(n) % 5
    st = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in state.items()}
    ref = mo.render(st, conf, rays, idx, True, True, noise)
    m._noise = {k: v.cuda() for k, v in noise.items()}
    out = m({k: v.cuda() for k, v in rays.items()}, idx.cuda(), if_pixel_input=True)
    for k in ('rgb_values', 'depth_values', 'normal_map', 'weights', 'sdf', 'grad_theta'):
        check(errlog, 'grid_variants', name, k, rel_err(out[k], ref[k]))
    names = [k for k, v in st.items() if v.requires_grad]
    g_o = dict(zip(names, torch.autograd.grad(mo.probe_loss(ref), [st[k] for k in names], allow_unused=True)))
    mo.probe_loss(out).backward()
    for k, p in m.named_parameters():
        if g_o.get(k) is not None:
            check(errlog, 'grid_variants', name, k, rel_err(p.grad, g_o[k]))


@pytest.mark.parametrize('P', [0, 1, 15, 16, 17, 63, 65])
def test_tiny_and_empty_point_sets(P):
    """Fewer points than a wave tile (16) / a workgroup (64), and none at all: the public evaluation methods."""
    from oracle import monosdf_oracle as mo
    conf, state, m = _build(64, 8, 'fp32')
    m.eval()
    g = torch.Generator().manual_seed(P + 1)
    x = (torch.rand(P, 3, generator=g) * 2 - 1)
    net = m.implicit_network
    with torch.no_grad():
        vals = net.get_sdf_vals(x.cuda())
    sdf, feat, grad = net.get_outputs(x.cuda())
    gu = net.gradient_sdf(x.cuda())
    assert vals.shape == (P, 1) and sdf.shape == (P, 1) and grad.shape == (P, 3) and gu.shape[0] == P
    assert feat.shape == (P, conf['feature_vector_size'])
    if P == 0:
        return
    st = {k: v.clone() for k, v in state.items()}
    sdf_o, feat_o, grad_o = mo.get_outputs(st, conf, x, create_graph=False)
    assert rel_err(sdf, sdf_o) < TOL and rel_err(feat, feat_o) < TOL and rel_err(grad, grad_o) < TOL
    assert rel_err(vals, mo.get_sdf_vals(st, conf, x)) < TOL


@pytest.mark.parametrize('P', [0, 1, 17, 65])
def test_tiny_and_empty_point_sets_grid_model(P):
    """The same on the hash-grid network (encoder node forms + level-major feature tensors): an empty point set has
    NULL tensors and must come back as empty outputs and an all-zero table gradient, not as an argument error."""
    from oracle import monosdf_oracle as mo
    from monosdf_amd.conf import ConfigTree
    from monosdf_amd.model.network import MonoSDFNetwork
    conf = config.grid_config(64, 0.1, 4, 2, 11, 8, 64)
    state = synth.make_state(conf, seed=5, jitter=0.3)
    m = MonoSDFNetwork(ConfigTree.from_dict(conf))
    m.load_state_dict({k: v.clone() for k, v in state.items()}, strict=True)
    m = m.cuda().train()
    net = m.implicit_network
    g = torch.Generator().manual_seed(P + 3)
    x = (torch.rand(P, 3, generator=g) * 2 - 1)
    with torch.no_grad():
        vals = net.get_sdf_vals(x.cuda())
    sdf, feat, grad = net.get_outputs(x.cuda())
    assert vals.shape == (P, 1) and sdf.shape == (P, 1) and grad.shape == (P, 3)
    assert feat.shape == (P, conf['feature_vector_size'])
    (sdf.sum() + feat.sum() * 0.1 + grad.sum()).backward()
    emb = net.encoding.embeddings
    assert emb.grad is not None and emb.grad.shape == emb.shape and torch.isfinite(emb.grad).all()
    if P == 0:
        assert float(emb.grad.abs().max()) == 0.0
        return
    st = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in state.items()}
    sdf_o, feat_o, grad_o = mo.get_outputs(st, conf, x)
    assert rel_err(sdf, sdf_o) < TOL and rel_err(feat, feat_o) < TOL and rel_err(grad, grad_o) < TOL
    assert rel_err(vals, mo.get_sdf_vals(st, conf, x)) < TOL
    (sdf_o.sum() + feat_o.sum() * 0.1 + grad_o.sum()).backward()
    assert rel_err(emb.grad, st['implicit_network.encoding.embeddings'].grad) < 2e-4


MODE_VARIANTS = {
    # (image mode, training, white background, per-image code, if_hdr)
    # uv input + training without options: tests/golden/mlp_w64_image_train.npz (recorded from the reference, round 4)
    'image_train_code_white': (True, True, True, True, False),
    'pixel_eval_white_hdr': (False, False, True, False, True),
    'pixel_train_code_hdr': (False, True, False, True, True),
    'image_eval_code': (True, False, False, True, False),
}


@pytest.mark.parametrize('name', sorted(MODE_VARIANTS))
def test_input_and_output_modes_against_oracle(name, errlog):
    """Combinations of uv / pixel input, train / eval, white background, per-image code and HDR output that no single
    golden holds together."""
    import numpy as np
    from oracle import monosdf_oracle as mo
    from monosdf_amd.conf import ConfigTree
    from monosdf_amd.model.network import MonoSDFNetwork
    image, training, white, code, hdr = MODE_VARIANTS[name]
    conf = config.mlp_config(64, 8)
    conf['white_bkgd'] = white
    conf['rendering_network']['per_image_code'] = code
    state = synth.make_state(conf, seed=51, jitter=0.3)
    n = 20
    if image:
        rng = np.random.default_rng(2)
        intr = torch.eye(4)[None].clone()
        intr[0, 0, 0], intr[0, 1, 1], intr[0, 0, 2], intr[0, 1, 2], intr[0, 0, 1] = 300., 310., 192., 190., 0.5
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        pose = torch.eye(4)[None].clone()
        pose[0, :3, :3] = torch.from_numpy(q).float()
        pose[0, :3, 3] = torch.tensor([0.1, -0.15, 0.05])
        inputs = {'uv': torch.from_numpy(rng.uniform(0, 384, size=(1, n, 2))).float(), 'pose': pose, 'intrinsics': intr}
        idx = torch.tensor([3])
    else:
        inputs = synth.make_rays(n, seed=7, random_pose=True)
        idx = torch.arange(n) % 7
    noise = synth.make_noise(conf, n, 128, seed=9) if training else None
    m = MonoSDFNetwork(ConfigTree.from_dict(conf), if_hdr=hdr)
    m.load_state_dict({k: v.clone() for k, v in state.items()}, strict=True)
    m = m.cuda().train(training)
    st = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in state.items()}
    ref = mo.render(st, conf, inputs, idx, not image, training, noise, if_hdr=hdr)
    m._noise = {k: v.cuda() for k, v in noise.items()} if noise else None
    out = m({k: v.cuda() for k, v in inputs.items()}, idx.cuda(), if_pixel_input=not image)
    assert set(out) == set(ref)
    for k in ref:
        assert out[k].shape == ref[k].shape, k
        check(errlog, 'modes', name, k, rel_err(out[k], ref[k]))
    if training:
        names = [k for k, v in st.items() if v.requires_grad]
        g_o = dict(zip(names, torch.autograd.grad(mo.probe_loss(ref), [st[k] for k in names], allow_unused=True)))
        mo.probe_loss(out).backward()
        for k, p in m.named_parameters():
            if g_o.get(k) is not None and p.grad is not None:
                check(errlog, 'modes', name, k, rel_err(p.grad, g_o[k]))
