"""The CPU restatement of the reference's MonoSDFLoss against fixtures recorded from the real class
(tests/golden/loss_*.npz, oracle/make_golden_loss.py): values and gradients, CPU only."""
import ast
import glob
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle as lo

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'loss_*.npz')))
GRAD_KEYS = ['rgb_values', 'depth_values', 'normal_map', 'grad_theta', 'grad_theta_nei']
SCALARS = ['loss', 'rgb_loss', 'eikonal_loss', 'smooth_loss', 'depth_loss', 'normal_l1', 'normal_cos']


def load_case(path):
    z = np.load(path)
    out = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('in.')}
    gt = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('gt.')}
    ref = {k[4:]: float(z[k]) for k in z.files if k.startswith('out.')}
    grads = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('grad.')}
    kw = ast.literal_eval(str(z['kw']))
    return out, gt, ref, grads, kw, int(z['step'])


def oracle_args(kw, step):
    weights = dict(eikonal=kw['eikonal_weight'], smooth=kw.get('smooth_weight', 0.005), depth=kw.get('depth_weight', 0.1),
                   normal_l1=kw.get('normal_l1_weight', 0.05), normal_cos=kw.get('normal_cos_weight', 0.05))
    return dict(weights=weights, step=step, end_step=kw.get('end_step', -1), if_gamma_loss=kw.get('if_gamma_loss', False),
                scale_invariant=kw.get('if_scale_invariant_depth', True))


def test_fixtures_present():
    assert len(GOLDEN) == 5


@pytest.mark.parametrize('path', GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_loss_oracle_matches_reference(path):
    out, gt, ref, grads, kw, step = load_case(path)
    leaves = {k: (v.clone().requires_grad_(True) if k in GRAD_KEYS else v) for k, v in out.items()}
    res = lo.monosdf_loss(leaves, gt, **oracle_args(kw, step))
    for k in SCALARS:
        assert abs(res[k].item() - ref[k]) <= 2e-6 * max(1.0, abs(ref[k])), k
    g = torch.autograd.grad(res['loss'], [leaves[k] for k in GRAD_KEYS], allow_unused=True)
    for k, gr in zip(GRAD_KEYS, g):
        gr = torch.zeros_like(out[k]) if gr is None else gr
        scale = max(float(grads[k].abs().max()), 1e-12)
        assert float((gr - grads[k]).abs().max()) <= 2e-5 * scale, k
