"""Training-trajectory parity with the reference (north_star: "PSNR within 0.1 dB of reference after equal
iterations").  tests/golden/traj_*.npz hold what the REFERENCE's model + MonoSDFLoss + Adam did over 200 steps
(oracle/make_golden_traj.py: training/monosdf_train.py:427-432, model/loss.py:252-311); here the HIP model + the
fused loss kernel + the same torch Adam replay the same steps: same initial weights, rays, targets and the same
six random draws per step (all from numpy seeds).  Compared: every loss term per step, beta, the sampler rounds,
the held-out PSNR at three checkpoints and the final parameters."""
import ast
import math

import numpy as np
import pytest
import torch

from helpers import check, digest
from oracle import config, synth

pytestmark = pytest.mark.gpu

TERMS = ('loss', 'rgb_loss', 'eikonal_loss', 'smooth_loss', 'depth_loss', 'normal_l1', 'normal_cos')


def _psnr(a, b):
    return -10.0 * math.log10(torch.mean((a - b) ** 2).item())


def _conf(spec):
    if spec.get('kind', 'mlp') == 'grid':
        g = spec['grid']
        return config.grid_config(spec['width'], spec['beta'], g['num_levels'], g['level_dim'], g['logmap'],
                                  g['base_size'], g['end_size'])
    return config.mlp_config(spec['width'], 8, spec['beta'])


# traj_grid_small: the reference's Python wiring over the restated hash kernels (parity of the hash arithmetic
# itself is unpinned: README) -- what it pins is the fused embedding scatter + Adam on the tables over 120 steps
# ... and the two MLP runs on the bf16x6 core as well (fp32-grade products), against the same bars
@pytest.mark.parametrize('precision', ['fp32', 'bf16x6'])
@pytest.mark.parametrize('name', ['traj_w64', 'traj_w64_sharp', 'traj_grid_small'])
def test_training_trajectory_matches_reference(name, precision, golden_dir, errlog):
    if precision != 'fp32' and name == 'traj_grid_small':
        pytest.skip('the hash-grid run is about the table scatter, not the matrix core')
    from monosdf_amd.conf import ConfigTree
    from monosdf_amd.model.loss import MonoSDFLoss
    from monosdf_amd.model.network import MonoSDFNetwork
    z = np.load('%s/%s.npz' % (golden_dir, name))
    spec = dict(ast.literal_eval(bytes(z['spec']).decode()))
    conf = _conf(spec)
    state = synth.make_state(conf, seed=spec['weight_seed'], jitter=spec['jitter'])
    model = MonoSDFNetwork(ConfigTree.from_dict(conf))
    model.load_state_dict(state, strict=True)
    model = model.cuda().set_precision(precision)
    loss_fn = MonoSDFLoss(rgb_loss='torch.nn.L1Loss', **spec['loss'])
    opt = torch.optim.Adam(model.parameters(), lr=spec['lr'])
    n = spec['n_rays']
    cuda = lambda d: {k: v.cuda() for k, v in d.items()}
    held = synth.make_rays(spec['held_out_rays'], seed=spec['held_out_seed'], random_pose=True)
    held_gt = synth.analytic_targets(held)['rgb'][0].cuda()
    held = cuda(held)
    idx = torch.arange(n).cuda()
    got = {k: [] for k in TERMS + ('beta', 'rounds')}
    psnr = []
    for step in range(spec['steps']):
        rays = synth.make_rays(n, seed=spec['ray_seed0'] + step, random_pose=True)
        gt = synth.analytic_targets(rays)
        model._noise = cuda(synth.make_noise_table(conf, n, seed=spec['noise_seed0'] + step))
        model.train()
        out = model(cuda(rays), idx, if_pixel_input=True)
        res = loss_fn(out, gt, if_pixel_input=True)
        opt.zero_grad()
        res['loss'].backward()
        opt.step()
        for k in TERMS:
            got[k].append(res[k].item())
        got['beta'].append(model.density.get_beta().item())
        got['rounds'].append(model.ray_sampler.last_rounds)
        if step + 1 in spec['checkpoints']:
            model.eval()
            model._noise = None
            with torch.no_grad():
                o = model(held, torch.arange(spec['held_out_rays']).cuda(), if_pixel_input=True)
            psnr.append(_psnr(o['rgb_values'], held_gt))
    # The yardstick: the fixture also holds four CONTROL runs of the reference itself, its initial weights perturbed by
    # one part in 1e6 -- what two equally exact implementations differ by after the same steps (training is chaotic:
    # at 200 steps the reference's own controls sit 0.08-0.16 dB from it).  north_star asks for 0.1 dB; where the
    # reference's own spread is larger than that, 1.5 x that spread is the bar.
    test = 'trajectory' if precision == 'fp32' else 'trajectory.' + precision
    ctrl_psnr = np.abs(z['control.psnr'] - z['psnr']).max(0)
    for i, (c, p, ref) in enumerate(zip(spec['checkpoints'], psnr, z['psnr'])):
        check(errlog, test, name, 'held-out PSNR after %d steps: |dB - reference| (controls: %.3f)' % (c, ctrl_psnr[i]),
              abs(p - ref), max(0.1, 1.5 * ctrl_psnr[i]))
    windows = ((0, 50), (50, 100), (100, spec['steps']))
    for k in ('loss', 'beta'):
        ref = z['traj.' + k]
        dev = np.abs(np.array(got[k]) - ref) / np.abs(ref).max()
        cdev = np.abs(z['control.' + k] - ref) / np.abs(ref).max()
        for lo, hi in windows:
            bar = 2.0 * cdev[:, lo:hi].max()
            check(errlog, test, name, '%s, steps %d-%d, relative to its maximum (2 x controls: %.1e)' % (k, lo + 1, hi, bar),
                  float(dev[lo:hi].max()), bar)
    # the other loss terms: recorded for the table
    for k in TERMS[1:]:
        ref = z['traj.' + k]
        dev = np.abs(np.array(got[k]) - ref) / np.abs(ref).max()
        errlog(test, name, k + ', all steps, relative to its maximum (not asserted)', float(dev.max()), float('inf'))
    # sampler rounds: a step whose max beta sits within rounding of beta0 needs one round more or fewer
    rounds_ref = z['traj.rounds']
    differ = float((np.array(got['rounds']) != rounds_ref).mean())
    cdiff = float((z['control.rounds'] != rounds_ref).mean(1).max())
    check(errlog, test, name, 'fraction of steps with a different round count (controls: %.2f)' % cdiff, differ,
          cdiff + 0.1)
    params = dict(model.named_parameters())
    worst = 0.0
    for key in z.files:
        if key.startswith('final.'):
            d, ref = digest(params[key[6:]]).numpy(), z[key]
            worst = max(worst, abs(d[1] - ref[1]) / (ref[1] + 1e-12))        # sum |w|
    errlog(test, name, 'final parameters: worst relative difference of sum|w| per tensor (not asserted)', worst,
           float('inf'))
