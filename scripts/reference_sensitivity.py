"""How far does the REFERENCE's own fp32 result move when the SDF values its sampler sees change in their last
bits?  (CPU only; the second yardstick of the parity table beside scripts/reference_conditioning.py)

Two correct fp32 implementations of the SDF network differ by ~1e-6 relative per value (two summation orders of a
256-term dot product; the HIP kernels against the reference: 1e-6 ... 5e-6 of max|sdf| in the stage tests).  The
error-bounded sampler turns such a difference into a different sample position wherever its inverse CDF is flat:
the LAST importance sample of a ray (column 95 of 98, u closest to 1) sits in the far tail of the cdf, where the
pdf is the 1e-5 floor (ray_sampler.py:191), and a relative change of 3e-7 of the SDF values moves it by up to 4e-5 of
the far bound on every 1-round golden case -- found with this script as the cause of the round-2 question about
`mlp_w64_hdr_eval` (z 3.7e-5 where fp32-vs-fp64 of the reference says 7.9e-7: the fp64 comparison holds only two
samples of that sensitivity).

For every golden case the oracle (bit-identical restatement of the reference on these cases) is run TRIALS times
in fp32 with the sampler's SDF values perturbed per value, everything else evaluated exactly as the reference does,
at the samples that come out.  Two perturbations, both of the size two correct fp32 SDF networks differ by:
  'perturbed_sdf'      s * (1 + EPS * U(-1, 1))           a RELATIVE last-bit change
  'perturbed_sdf_abs'  s + EPS * max|s| * U(-1, 1)        an ABSOLUTE change of EPS of the batch's largest |sdf| -- what the
                       HIP kernels measure against the reference in the stage tests (sdf_stages.fp32: 5e-7 ... 9e-7 of
                       max|sdf|, the rounding of 256-term sums whose terms are O(max|sdf|) whatever the result).  Near the
                       surface |s| << max|s|, so the relative form understates it by orders of magnitude exactly where the
                       sampler puts its samples once beta is small (the 4- and 5-round cases): found as the cause of the
                       round-2 question about the weight gradients of `mlp_w64_train_k5nc` (3-5 x the relative yardstick,
                       1.2-1.8 x this one).  Per output tensor and per parameter gradient: the
largest deviation from the unperturbed run, as a fraction of the tensor's max-abs (z as a fraction of far).
Writes profiles/r03_reference_sensitivity.json (merged with the fp32-vs-fp64 figures of
profiles/r02_reference_conditioning.json as 'fp64')."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from helpers import ALL_CASES, Case, rel_err               # noqa: E402
from oracle import monosdf_oracle as mo                    # noqa: E402

EPS = 1e-6
TRIALS = 16


def run(c, gen=None, absolute=False):
    state = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in c.state.items()}
    z_override = None
    if gen is not None:
        def fn(p):
            s = mo.get_sdf_vals(c.state, c.conf, p)
            u = 2 * torch.rand(s.shape, generator=gen) - 1
            return s + EPS * s.abs().max() * u if absolute else s * (1 + EPS * u)
        if c.pixel:
            dirs, cam = c.inputs['ray_dirs'], c.inputs['ray_cam_loc']
        else:        # image mode (round 4): the rays render() forms from uv / pose / intrinsics
            dirs, cam = mo.camera_rays(c.inputs['uv'], c.inputs['pose'], c.inputs['intrinsics'])
            cam = cam.unsqueeze(1).repeat(1, dirs.shape[1], 1).reshape(-1, 3)
            dirs = dirs.reshape(-1, 3)
        with torch.no_grad():
            z_override = mo.error_bound_sampler(c.state, c.conf, dirs, cam, c.training, c.noise, sdf_fn=fn)
    out = mo.render(state, c.conf, c.inputs, c.indices, c.pixel, c.training, c.noise, z_override=z_override,
                    if_hdr=c.spec.get('if_hdr', False))
    grads = {}
    if c.training:
        names = [n for n in state if state[n].requires_grad]
        gs = torch.autograd.grad(mo.probe_loss(out), [state[n] for n in names], allow_unused=True)
        grads = {n: g for n, g in zip(names, gs) if g is not None}
    return {k: v.detach() for k, v in out.items()}, grads


def main():
    """No argument: every pixel-mode golden case -> profiles/r03_reference_sensitivity.json (round 3's run).
    `--only case [case ...]` (round 4): the named cases (image-mode ones too), their fp32-vs-fp64 figures computed here
    with scripts/reference_conditioning.run, merged with round 3's file into profiles/r04_reference_sensitivity.json."""
    only = sys.argv[sys.argv.index('--only') + 1:] if '--only' in sys.argv else None
    fp64_path = os.path.join(ROOT, 'profiles', 'r02_reference_conditioning.json')
    fp64 = json.load(open(fp64_path)) if os.path.exists(fp64_path) else {}
    res = {'eps': EPS, 'trials': TRIALS, 'cases': {}}
    out_path = os.path.join(ROOT, 'profiles', 'r03_reference_sensitivity.json')
    if only:
        import reference_conditioning as rc
        res = json.load(open(out_path))
        out_path = os.path.join(ROOT, 'profiles', 'r04_reference_sensitivity.json')
        if os.path.exists(out_path):
            res = json.load(open(out_path))
        for name in only:
            c = Case(name)
            o32, g32 = rc.run(c, torch.float32)
            ent = {}
            for variant in ((False, True) if c.pixel else (False,)):
                o64, g64 = rc.run(c, torch.float64, variant)
                for k in o32:
                    ent['out.' + k] = max(ent.get('out.' + k, 0.0), rel_err(o32[k], o64[k]))
                for k in g32:
                    ent['grad.' + k] = max(ent.get('grad.' + k, 0.0), rel_err(g32[k], g64[k]))
            fp64[name] = ent
    for name in (only or ALL_CASES):
        c = Case(name)
        if not c.pixel and not only:
            continue              # round 3: image-mode cases were left to their pixel twins
        o0, g0 = run(c)
        both = {}
        for absolute in (False, True):
            ent = {}
            gen = torch.Generator().manual_seed(0)
            for _ in range(TRIALS):
                o1, g1 = run(c, gen, absolute)
                for k in o0:
                    if o0[k].shape != o1[k].shape:
                        continue
                    e = (o1[k] - o0[k]).abs().max().item() / 3.85 if k in ('z_vals',) else rel_err(o1[k], o0[k])
                    ent['out.' + k] = max(ent.get('out.' + k, 0.0), e)
                for k in g0:
                    ent['grad.' + k] = max(ent.get('grad.' + k, 0.0), rel_err(g1[k], g0[k]))
            both['perturbed_sdf_abs' if absolute else 'perturbed_sdf'] = ent
        ent = both['perturbed_sdf_abs']
        res['cases'][name] = dict(both, fp64=fp64.get(name, {}))
        worst = sorted(ent.items(), key=lambda kv: -kv[1])[:3]
        print('%-24s %s' % (name, ', '.join('%s %.1e' % kv for kv in worst)), flush=True)
    json.dump(res, open(out_path, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
