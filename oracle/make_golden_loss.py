"""Records the REAL reference MonoSDFLoss (code/model/loss.py) on seeded synthetic inputs into
tests/golden/loss_*.npz: inputs, the seven output scalars and the gradients of `loss` with respect to
every model output.  Container-only (needs /root/reference); the fixtures are what travels.

    python -m oracle.make_golden_loss
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

from . import ref_loader

OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')

# name -> (N rays, S samples, ctor kwargs, step, mask density, degenerate depth)
CASES = {
    'loss_default': dict(n=96, s=12, kw=dict(eikonal_weight=0.05), step=0),
    'loss_weights_decay': dict(n=200, s=9, kw=dict(eikonal_weight=0.1, smooth_weight=0.01, depth_weight=0.2,
                                                    normal_l1_weight=0.03, normal_cos_weight=0.07, end_step=1000),
                               step=137),
    'loss_gamma_plain_depth': dict(n=64, s=7, kw=dict(eikonal_weight=0.05, if_gamma_loss=True,
                                                       if_scale_invariant_depth=False), step=0),
    'loss_empty_mask': dict(n=40, s=6, kw=dict(eikonal_weight=0.05), step=0, mask_p=2.0),
    'loss_constant_depth': dict(n=48, s=6, kw=dict(eikonal_weight=0.05), step=0, const_depth=True),
}


def load_reference_loss():
    ref_loader.load()
    tv, tr = types.ModuleType('torchvision'), types.ModuleType('torchvision.transforms')
    tr.ToPILImage = lambda *a, **k: None            # utils/general.py:67 builds one at import time
    tv.transforms = tr
    sys.modules.setdefault('torchvision', tv)
    sys.modules.setdefault('torchvision.transforms', tr)
    import model.loss as ref_loss                   # noqa: E402  (reference module)
    return ref_loss


def make_io(spec, seed):
    g = torch.Generator().manual_seed(seed)
    n, s = spec['n'], spec['s']
    r = lambda *shape: torch.rand(*shape, generator=g)
    rn = lambda *shape: torch.randn(*shape, generator=g)
    out = {'rgb_values': r(n, 3) * 1.1 - 0.05, 'depth_values': r(n, 1) * 2 + 0.5, 'normal_map': rn(n, 3) * 0.7,
           'sdf': rn(n, s) + 0.3, 'grad_theta': rn(2 * n, 3) * 0.8, 'grad_theta_nei': rn(2 * n, 3) * 0.8}
    if spec.get('const_depth'):
        out['depth_values'] = torch.full((n, 1), 1.25)          # singular 2x2 system -> scale = shift = 0
    gt = {'rgb': r(1, n, 3), 'depth': r(1, n, 1) * 0.04, 'normal': rn(1, n, 3),
          'mask': (r(1, n, 1) > spec.get('mask_p', 0.3)).float()}
    return out, gt


def main():
    ref_loss = load_reference_loss()
    os.makedirs(OUT_DIR, exist_ok=True)
    for i, (name, spec) in enumerate(CASES.items()):
        out, gt = make_io(spec, 100 + i)
        grad_keys = [k for k in out if k != 'sdf']
        leaves = {k: (v.clone().requires_grad_(True) if k in grad_keys else v) for k, v in out.items()}
        with contextlib.redirect_stdout(io.StringIO()):         # the reference prints per step (loss.py:164,203)
            mod = ref_loss.MonoSDFLoss(rgb_loss='torch.nn.L1Loss', **spec['kw'])
            mod.step = spec['step']
            res = mod(leaves, gt, if_pixel_input=True)
        loss = res['loss']
        grads = torch.autograd.grad(loss, [leaves[k] for k in grad_keys], allow_unused=True)
        rec = {'in.' + k: v.numpy() for k, v in out.items()}
        rec.update({'gt.' + k: v.numpy() for k, v in gt.items()})
        rec.update({'out.' + k: np.float64(v.item()) for k, v in res.items()})
        for k, gr in zip(grad_keys, grads):
            rec['grad.' + k] = (torch.zeros_like(out[k]) if gr is None else gr).numpy()
        rec['step'] = np.int64(spec['step'])
        rec['kw'] = np.array(repr(spec['kw']))
        path = os.path.join(OUT_DIR, name + '.npz')
        np.savez_compressed(path, **rec)
        print('%-26s loss=%.6f  %5.1f KB' % (name, float(loss), os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
