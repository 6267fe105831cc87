"""Training-trajectory fixture recorded from the REAL reference (container-only, TEST INFRASTRUCTURE):

    python -m oracle.make_golden_traj

The reference's MonoSDFNetwork (width 64) is trained for STEPS Adam steps with the reference's MonoSDFLoss
(training/monosdf_train.py:403-485 without the data loader: model -> loss -> backward -> optimizer.step, lr and
betas of the fork's confs; model/loss.py:252-311) on a closed-form scene (oracle/synth.analytic_targets).  Rays and
all six random draws of every step come from numpy seeds (oracle/synth.py), injected into the reference through
ref_loader.inject_rng, so the fixture holds only the trajectory: the loss terms, beta and the sampler rounds per
step, digests of the final parameters, and the PSNR on held-out rays at three checkpoints (utils/rend_util.py:17-24).
tests/test_gpu_trajectory.py replays the same steps with the HIP model + fused loss."""
import contextlib
import io
import math
import os

import numpy as np
import torch

from . import config, make_golden_loss, ref_loader, synth

OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
BASE = dict(width=64, weight_seed=11, jitter=0.0, n_rays=64, steps=200, lr=5.0e-4, ray_seed0=1000, noise_seed0=3000,
            held_out_seed=77, held_out_rays=256, checkpoints=(50, 100, 200), beta=0.02,
            loss=dict(eikonal_weight=0.05, smooth_weight=0.005, depth_weight=0.1, normal_l1_weight=0.05,
                      normal_cos_weight=0.05))
# two runs: the confs' initial beta (the sampler converges in one round throughout), and a sharp start where beta
# falls to 6e-4 and the sampler needs 2 to 4 rounds per step
SPECS = {'traj_w64': dict(BASE, beta=0.1), 'traj_w64_sharp': dict(BASE, beta=0.02),
         # the hash-grid model (4 levels, 2^10 entries: the reference's Python wiring over the restated kernels of
         # ref_loader.FakeHashBackend -- pins the wiring and Adam on the embeddings, not the kernel arithmetic)
         'traj_grid_small': dict(BASE, beta=0.1, kind='grid', steps=120, checkpoints=(40, 80, 120),
                                 grid=dict(num_levels=4, level_dim=2, logmap=10, base_size=16, end_size=64))}


def conf_of(spec):
    if spec.get('kind', 'mlp') == 'grid':
        g = spec['grid']
        return config.grid_config(spec['width'], spec['beta'], g['num_levels'], g['level_dim'], g['logmap'],
                                  g['base_size'], g['end_size'])
    return config.mlp_config(spec['width'], 8, spec['beta'])
TERMS = ('loss', 'rgb_loss', 'eikonal_loss', 'smooth_loss', 'depth_loss', 'normal_l1', 'normal_cos')


def digest(t):
    a = t.detach().double().flatten()
    idx = torch.linspace(0, a.numel() - 1, min(16, a.numel())).long()
    return np.concatenate([[a.sum().item(), a.abs().sum().item(), (a * a).sum().item()], a[idx].numpy()])


def psnr(a, b):
    """utils/rend_util.get_psnr (17-24)."""
    return (-10.0 * torch.log(torch.mean((a - b) ** 2)) / math.log(10.0)).item()


def run(name, spec, perturb=0.0, perturb_seed=9):
    """perturb > 0: a control run -- the same reference, every initial weight multiplied by 1 + perturb * N(0,1):
    the spread two equally exact implementations show after the same steps."""
    ref_loss = make_golden_loss.load_reference_loss()
    conf = conf_of(spec)
    state = synth.make_state(conf, seed=spec['weight_seed'], jitter=spec['jitter'])
    if perturb > 0:
        g = torch.Generator().manual_seed(perturb_seed)
        state = {k: (v * (1.0 + perturb * torch.randn(v.shape, generator=g)) if v.dtype.is_floating_point else v)
                 for k, v in state.items()}
    model = ref_loader.build_model(conf, state, training=True)
    n = spec['n_rays']
    with contextlib.redirect_stdout(io.StringIO()):
        loss_fn = ref_loss.MonoSDFLoss(rgb_loss='torch.nn.L1Loss', **spec['loss'])
    opt = torch.optim.Adam(model.parameters(), lr=spec['lr'])
    held = synth.make_rays(spec['held_out_rays'], seed=spec['held_out_seed'], random_pose=True)
    held_gt = synth.analytic_targets(held)
    rounds = [0]
    orig = model.implicit_network.get_sdf_vals

    def counted(x):
        rounds[0] += 1
        return orig(x)
    model.implicit_network.get_sdf_vals = counted
    rec = {k: [] for k in TERMS + ('beta', 'rounds')}
    ck = {}
    for step in range(spec['steps']):
        rays = synth.make_rays(n, seed=spec['ray_seed0'] + step, random_pose=True)
        gt = synth.analytic_targets(rays)
        noise = synth.make_noise_table(conf, n, seed=spec['noise_seed0'] + step)
        model.train()
        rounds[0] = 0
        with ref_loader.inject_rng(noise, n, conf), contextlib.redirect_stdout(io.StringIO()):
            out = model(rays, torch.arange(n), if_pixel_input=True)
            res = loss_fn(out, gt, if_pixel_input=True)
        opt.zero_grad()
        res['loss'].backward()
        opt.step()
        for k in TERMS:
            rec[k].append(float(res[k]))
        rec['beta'].append(float(model.density.get_beta()))
        rec['rounds'].append(rounds[0])
        if step + 1 in spec['checkpoints']:
            model.eval()
            o = model(held, torch.arange(spec['held_out_rays']), if_pixel_input=True)
            ck[step + 1] = psnr(o['rgb_values'].detach(), held_gt['rgb'][0])
            print('step %4d  loss %.5f  beta %.5f  rounds %d  held-out PSNR %.3f dB' %
                  (step + 1, rec['loss'][-1], rec['beta'][-1], rec['rounds'][-1], ck[step + 1]))
    if perturb > 0:
        return {'psnr': np.array([ck[c] for c in spec['checkpoints']]), 'loss': np.array(rec['loss']),
                'rounds': np.array(rec['rounds']), 'beta': np.array(rec['beta'])}
    controls = [run(name, spec, perturb=1e-6, perturb_seed=9 + i) for i in range(4)]
    data = {'spec': np.frombuffer(repr(sorted(spec.items())).encode(), dtype=np.uint8),
            'psnr': np.array([ck[c] for c in spec['checkpoints']])}
    data.update({'control.' + k: np.stack([c[k] for c in controls]) for k in controls[0]})
    data.update({'traj.' + k: np.array(v) for k, v in rec.items()})
    for pname, p in model.named_parameters():
        data['final.' + pname] = digest(p)
    out = os.path.join(OUT_DIR, name + '.npz')
    np.savez_compressed(out, **data)
    print('%s %.1f KB' % (name, os.path.getsize(out) / 1024))


def main(argv=()):
    for name, spec in SPECS.items():
        if argv and name not in argv:
            continue
        run(name, spec)


if __name__ == '__main__':
    import sys
    main(sys.argv[1:])
