"""The CPU oracle against the golden vectors recorded from the real reference.

Pins oracle/monosdf_oracle.py (and the hash-grid *wiring*) to the reference's own
outputs: tests/golden/*.npz were written by oracle/make_golden.py importing
/root/reference on CPU.  fp32; tolerance 2e-4 relative to each tensor's max
(the sampler's inverse CDF amplifies last-bit differences in cumsum).
"""
import numpy as np
import pytest
import torch

from oracle import monosdf_oracle as mo
from oracle import config, synth
from helpers import ALL_CASES, Case, digest, rel_err

TOL = 2e-4


@pytest.mark.parametrize('name', ALL_CASES)
def test_forward_and_grads_match_reference(name):
    c = Case(name)
    state = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in c.state.items()}
    trace = {}
    out = mo.render(state, c.conf, c.inputs, c.indices, c.pixel, c.training, c.noise,
                    if_hdr=c.spec.get('if_hdr', False))
    assert set(out) == set(c.out)
    for k, ref in c.out.items():
        assert out[k].shape == ref.shape, k
        assert rel_err(out[k], ref) < TOL, (k, rel_err(out[k], ref))
    if c.grads or c.gdig:
        loss = mo.probe_loss(out)
        assert abs(loss.item() - c.loss) < 1e-5 * max(1.0, abs(c.loss))
        names = list(c.grads) + list(c.gdig)
        grads = torch.autograd.grad(loss, [state[n] for n in names], allow_unused=True)
        for n, g in zip(names, grads):
            assert g is not None, n
            if n in c.grads:
                assert rel_err(g, c.grads[n]) < 5e-4, (n, rel_err(g, c.grads[n]))
            else:
                d, ref = digest(g), c.gdig[n].double()
                scale = ref[1] / max(1, g.numel()) + 1e-12       # mean |g|
                assert abs(d[0] - ref[0]) < 5e-4 * ref[1] + 1e-9, n
                assert ((d[3:] - ref[3:]).abs().max() < 5e-2 * scale * 16 + 5e-4 * ref[3:].abs().max()), n
            if n in c.gtab:
                from oracle import synth
                got, ref = synth.table_fingerprint(g, c.state[n.replace('embeddings', 'offsets')].numpy()), c.gtab[n]
                for k in ('level_abs', 'level_sq', 'proj', 'top_val'):
                    assert np.abs(got[k] - ref[k].numpy()).max() <= 1e-5 * np.abs(ref[k].numpy()).max(), (n, k)
                assert np.array_equal(got['top_idx'], ref['top_idx'].numpy()), n


def test_sampler_round_counts():
    seen = set()
    for name in ['mlp_w64_eval', 'mlp_w64_eval_sharp', 'mlp_w64_eval_vsharp', 'mlp_w64_eval_k3',
                 'mlp_w64_eval_k4', 'mlp_w64_eval_k5', 'mlp_w64_eval_k5nc', 'mlp_w64_train_k4',
                 'mlp_w64_train_k5nc']:
        c = Case(name)
        trace = {}
        rays = c.inputs
        mo.error_bound_sampler(c.state, c.conf, rays['ray_dirs'], rays['ray_cam_loc'], c.training, c.noise,
                               trace=trace)
        assert trace['rounds'] == c.rounds
        beta0 = mo.get_beta(c.state, c.conf)
        assert bool(trace['beta'].max() <= beta0) == c.converged
        seen.add((c.rounds, c.converged))
    # the fixtures reach every exit of the loop: 1..5 rounds converged, and max_total_iters without converging
    assert {(k, True) for k in range(1, 6)} | {(5, False)} <= seen


@pytest.mark.parametrize('name', ['mlp_w64_eval_k2_trace', 'mlp_w64_eval_k4', 'mlp_w64_eval_k5',
                                  'mlp_w64_eval_k5nc', 'mlp_w64_train_k4'])
def test_sampler_rounds_match_reference_intermediates(name):
    """Per round: sorted z, merged sdf, d*, beta after the bisection, cdf, u, new samples -- recorded inside the
    reference's get_z_vals (oracle/ref_loader.record_sampler) -- against the oracle's."""
    c = Case(name)
    trace = {}
    rays = c.inputs
    mo.error_bound_sampler(c.state, c.conf, rays['ray_dirs'], rays['ray_cam_loc'], c.training, c.noise,
                           trace=trace)
    assert len(trace['per_round']) == len(c.trace) == c.rounds
    for got, ref in zip(trace['per_round'], c.trace):
        for k in ('z', 'sdf', 'dstar', 'beta', 'cdf', 'u', 'samples'):
            assert got[k].shape == ref[k].shape, k
            assert torch.allclose(got[k], ref[k], rtol=1e-5, atol=1e-6), (k, (got[k] - ref[k]).abs().max())


def test_stage_vectors(golden_dir):
    z = np.load(golden_dir + '/stages.npz')
    t = lambda k: torch.from_numpy(z[k])
    conf = config.mlp_config(64, 8)
    state = synth.make_state(conf, seed=3, jitter=0.3)
    zz, _, far = mo.uniform_z(conf, t('uni.d'), t('uni.o'), 128)
    assert torch.equal(zz, t('uni.z_eval')) and torch.equal(far, t('uni.far'))
    zt, _, _ = mo.uniform_z(conf, t('uni.d'), t('uni.o'), 128, t('uni.jitter'))
    assert torch.allclose(zt, t('uni.z_train'), rtol=0, atol=1e-6)
    near, far = mo.far_from_cube(t('cube.o'), t('cube.d'), 1.1, 0.0, mo.sampler_far(conf))
    assert torch.equal(near, t('cube.near')) and torch.equal(far, t('cube.far'))
    s, b = t('dens.sdf'), t('dens.beta')
    assert torch.allclose(mo.laplace_density(s, torch.tensor(0.1 + 1e-4)), t('dens.scalar'), rtol=1e-6, atol=1e-6)
    assert torch.allclose(mo.laplace_density(s, b), t('dens.perray'), rtol=1e-6, atol=1e-6)
    zq = t('eb.z')
    eb = mo.error_bound(b, s, zq[:, 1:] - zq[:, :-1], t('eb.dstar'))
    assert torch.allclose(eb, t('eb.out'), rtol=1e-5, atol=1e-7)
    st01 = dict(state)
    st01['density.beta'] = torch.tensor(0.1)
    w = mo.volume_rendering(st01, conf, zq, s.reshape(-1, 1))
    assert torch.allclose(w, t('vr.weights'), rtol=1e-5, atol=1e-7)
    assert torch.equal(mo.positional_encoding(t('pe.x'), 6), t('pe.out6'))
    assert torch.equal(mo.positional_encoding(t('pe.x'), 4), t('pe.out4'))
    pts = t('net.pts')
    sdf, feat, grad = mo.get_outputs(state, conf, pts)
    assert rel_err(sdf, t('net.sdf')) < 1e-5 and rel_err(feat, t('net.feat')) < 1e-5
    assert rel_err(grad, t('net.grad')) < 1e-5
    assert rel_err(mo.get_sdf_vals(state, conf, pts), t('net.sdf_vals')) < 1e-5
    assert rel_err(mo.gradient_sdf(state, conf, pts), t('net.grad_unclamped')) < 1e-5
    rgb = mo.color_network(state, conf, pts, t('net.grad'), t('col.dirs'), t('net.feat'))
    assert rel_err(rgb, t('col.rgb')) < 1e-5


def volume_fixture(golden_dir):
    import ast
    z = np.load(golden_dir + '/volume_w64_128.npz')
    spec = dict(ast.literal_eval(bytes(z['spec']).decode()))
    conf = config.mlp_config(spec['width'], 8)
    state = synth.make_state(conf, seed=spec['weight_seed'], jitter=spec['jitter'])
    return z, spec, conf, state


def volume_moments(a):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum()])


def test_sdf_volume_block_matches_reference_get_surface_sliding(golden_dir):
    """the oracle's restatement of plots.get_surface_sliding's SDF loop against the volume the
    REFERENCE handed to marching cubes (oracle/make_golden_volume.py; plots.py:108-205)."""
    z, spec, conf, state = volume_fixture(golden_dir)
    lo, hi = spec['grid_boundary']
    with torch.no_grad():
        vol = mo.sdf_volume_block(lambda p: mo.sdf_network_raw(state, conf, p)[:, 0], (lo,) * 3, (hi,) * 3,
                                  spec['resolution']).numpy()
    k = spec['stride']
    assert np.array_equal(vol[::k, ::k, ::k], z['sub'])
    assert np.array_equal(vol[:, :, vol.shape[0] // 2], z['plane'])
    assert np.allclose(volume_moments(vol), z['moments'], rtol=1e-12)
    thr = 2 * (hi - lo) / spec['resolution'] * 8 / 8
    assert int((np.abs(vol) < thr).sum()) == int(z['near_count'])


def plumbing_fixture(golden_dir):
    import ast
    z = np.load(golden_dir + '/plumbing_uniform64.npz')
    spec = dict(ast.literal_eval(bytes(z['spec']).decode()))
    conf = config.mlp_config(spec['width'], 8)
    state = synth.make_state(conf, seed=spec['weight_seed'], jitter=spec['jitter'])
    rays = synth.make_rays(spec['n_rays'], seed=spec['ray_seed'], random_pose=True)
    return z, spec, conf, state, rays


def test_configs0_uniform_sampler_plumbing(golden_dir):
    """BASELINE.json configs[0]: 512 rays x 64 uniform samples, fp32 on the CPU -- the oracle's composition against
    the reference's own UniformSampler + networks + volume_rendering (oracle/make_golden.py: run_plumbing)."""
    z, spec, conf, state, rays = plumbing_fixture(golden_dir)
    n = spec['n_rays']
    out = mo.render_uniform(state, conf, rays, torch.arange(n), spec['n_samples'])
    t = lambda k: torch.from_numpy(z[k])
    assert out['z_vals'].shape == (512, 64) and torch.equal(out['z_vals'], t('out.z_vals'))
    for k in ('rgb_values', 'depth_values', 'normal_map'):
        assert rel_err(out[k], t('out.' + k)) < 1e-5, (k, rel_err(out[k], t('out.' + k)))
    for k in ('sdf', 'weights', 'rgb'):
        assert rel_err(out[k][::8], t('sub.' + k)) < 1e-5, k
