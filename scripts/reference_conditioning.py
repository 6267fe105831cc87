"""How far is the REFERENCE's fp32 result from exact arithmetic, per golden case and per tensor?  (CPU only)

For every golden case the oracle (bit-identical restatement of the reference on these cases,
tests/test_oracle_golden.py) is evaluated in fp32 and in fp64 on the same weights, rays and random draws:
outputs and the gradients of the probe loss.  |fp32 - fp64| / max|fp64| per tensor is the yardstick for
profiles/r02_parity_errors.md: an implementation that is as exact as the reference differs from it by about this
much wherever the problem amplifies rounding (flat stretches of the sampler's cdf, near-step densities at small
beta).  Writes profiles/r02_reference_conditioning.json."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from helpers import ALL_CASES, Case, rel_err               # noqa: E402
from oracle import monosdf_oracle as mo                    # noqa: E402


def run(c, dtype, sampler_sdf_fp32=False):
    """sampler_sdf_fp32: the fp64 run's sampler sees the fp32 network's SDF values, so that only the sampler's own
    arithmetic changes precision (whether a sample falls on a flat stretch of the cdf depends on the last bits
    of its inputs: two exact variants bracket the spread better than one)."""
    cast = lambda v: v.to(dtype) if v.dtype.is_floating_point else v
    state = {k: cast(v).clone().requires_grad_(v.dtype.is_floating_point) for k, v in c.state.items()}
    inputs = {k: cast(v) for k, v in c.inputs.items()}
    noise = {k: cast(v) for k, v in c.noise.items()} if c.noise else None
    z_override = None
    if sampler_sdf_fp32 and c.pixel:
        fn = lambda p: mo.get_sdf_vals(c.state, c.conf, p.float()).to(dtype)
        with torch.no_grad():
            z_override = mo.error_bound_sampler(state, c.conf, inputs['ray_dirs'], inputs['ray_cam_loc'], c.training,
                                                noise, sdf_fn=fn)
    out = mo.render(state, c.conf, inputs, c.indices, c.pixel, c.training, noise, z_override=z_override,
                    if_hdr=c.spec.get('if_hdr', False))
    grads = {}
    if c.training:
        names = [n for n in state if state[n].requires_grad]
        gs = torch.autograd.grad(mo.probe_loss(out), [state[n] for n in names], allow_unused=True)
        grads = {n: g for n, g in zip(names, gs) if g is not None}
    return out, grads


def main():
    res = {}
    for name in ALL_CASES:
        c = Case(name)
        if c.conf.get('Grid_MLP', False) and c.conf['implicit_network'].get('use_grid_feature', True):
            continue                               # the hash-grid restatement is fp32 only
        o32, g32 = run(c, torch.float32)
        ent = {}
        for variant in (False, True):
            o64, g64 = run(c, torch.float64, variant)
            for k in o32:
                ent['out.' + k] = max(ent.get('out.' + k, 0.0), rel_err(o32[k], o64[k]))
            for k in g32:
                ent['grad.' + k] = max(ent.get('grad.' + k, 0.0), rel_err(g32[k], g64[k]))
        res[name] = ent
        worst = sorted(ent.items(), key=lambda kv: -kv[1])[:3]
        print('%-24s %s' % (name, ', '.join('%s %.1e' % kv for kv in worst)))
    json.dump(res, open(os.path.join(ROOT, 'profiles', 'r02_reference_conditioning.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
