"""Shared helpers for the parity tests (load a golden case, rebuild its inputs)."""
import ast
import os

import numpy as np
import torch

from oracle import config, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def conf_from_spec(spec):
    kind = spec['kind']
    if kind == 'mlp':
        c = config.mlp_config(spec['width'], spec.get('depth', 8), spec.get('beta', 0.1))
    elif kind == 'gridless':
        c = config.gridless_config(spec['width'], spec.get('depth', 8), spec.get('beta', 0.1))
    else:
        c = config.grid_config(spec['width'], spec.get('beta', 0.1), spec.get('num_levels', 16),
                               spec.get('level_dim', 2), spec.get('logmap', 19),
                               spec.get('base_size', 16), spec.get('end_size', 2048))
    if spec.get('white_bkgd', False):
        c['white_bkgd'] = True
    if spec.get('per_image_code', False):
        c['rendering_network']['per_image_code'] = True
    return c


class Case:
    """One tests/golden/<name>.npz: spec, rebuilt conf/state, recorded inputs/noise/outputs."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + '.npz'))
        self.name = name
        self.spec = dict(ast.literal_eval(bytes(z['spec']).decode()))
        self.conf = conf_from_spec(self.spec)
        self.state = synth.make_state(self.conf, seed=self.spec.get('weight_seed', 0),
                                      jitter=self.spec['jitter'])
        self.rounds = int(z['rounds'])
        self.indices = torch.from_numpy(z['indices'])
        self.training = self.spec['training']
        self.pixel = not self.spec.get('image_mode', False)
        pick = lambda p: {k[len(p):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(p)}
        self.inputs, self.noise, self.out = pick('in.'), pick('noise.'), pick('out.')
        self.grads, self.gdig = pick('grad.'), pick('gdig.')
        self.loss = float(z['loss']) if 'loss' in z.files else None


def digest(t):
    a = t.detach().double().flatten().cpu()
    idx = torch.linspace(0, a.numel() - 1, min(16, a.numel())).long()
    return torch.cat([torch.stack([a.sum(), a.abs().sum(), (a * a).sum()]), a[idx]])


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


ALL_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN)
                   if f.endswith('.npz') and f != 'stages.npz' and not f.startswith('loss_'))
