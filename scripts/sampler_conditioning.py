"""How far does the REFERENCE's own fp32 sampler sit from exact arithmetic?  (container or GPU box, CPU only)

For every eval-mode golden case: the oracle sampler (bit-identical to the reference on these cases,
tests/test_oracle_golden.py) in fp32 against the same algorithm in fp64, as a fraction of the far bound 3.85 --
the yardstick for the z_vals rows of profiles/r02_parity_errors.md: where the inverse CDF lands on a flat stretch
of the cdf, a last-bit difference of a prefix sum moves a sample by up to one interval, in the reference too."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from helpers import ALL_CASES, Case                      # noqa: E402
from oracle import monosdf_oracle as mo                  # noqa: E402


def main():
    out = {}
    for name in ALL_CASES:
        c = Case(name)
        if c.training or not c.pixel:
            continue
        rays = c.inputs
        t32, t64 = {}, {}
        z32, _ = mo.error_bound_sampler(c.state, c.conf, rays['ray_dirs'], rays['ray_cam_loc'], False, None, trace=t32)
        st64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in c.state.items()}
        # same SDF values in both runs (the fp32 network): only the sampler's own arithmetic changes precision
        sdf32 = lambda p: mo.get_sdf_vals(c.state, c.conf, p.float()).double()
        with torch.no_grad():
            z64, _ = mo.error_bound_sampler(st64, c.conf, rays['ray_dirs'].double(), rays['ray_cam_loc'].double(), False,
                                            None, sdf_fn=sdf32, trace=t64)
        err = ((z32.double() - z64).abs().max() / 3.85).item() if t32['rounds'] == t64['rounds'] else float('nan')
        out[name] = {'rounds_fp32': t32['rounds'], 'rounds_fp64': t64['rounds'], 'z_fp32_vs_fp64': err,
                     'z_fp32_vs_reference': ((z32 - c.out['z_vals']).abs().max() / 3.85).item()}
        print('%-24s rounds %d/%d  |z32 - z64| / far = %.2e' % (name, t32['rounds'], t64['rounds'], err))
    path = os.path.join(ROOT, 'profiles', 'r02_sampler_conditioning.json')
    json.dump(out, open(path, 'w'), indent=1)


if __name__ == '__main__':
    main()
