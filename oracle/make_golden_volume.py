"""Fixture for the coarse-to-fine SDF volume evaluator, recorded from the REAL reference
(container-only, TEST INFRASTRUCTURE): ``python -m oracle.make_golden_volume``.

Runs the reference's ``utils.plots.get_surface_sliding`` (utils/plots.py:108-226) at resolution 128 on the
reference's own ImplicitNetwork (width 64, weights from oracle/synth.py) with skimage / trimesh stubbed, and keeps
the volume it hands to ``measure.marching_cubes`` (plots.py:199-205).  The 128^3 float32 volume is 8 MB, so the
fixture stores every 4th voxel per axis (32^3 = 131 KB) plus three moments of the whole array and of the set of
voxels the pyramid refined down to the finest level (|sdf| below the last threshold)."""
import contextlib
import io
import os

import numpy as np
import torch

from . import config, ref_loader, synth

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden',
                   'volume_w64_128.npz')
SPEC = dict(width=64, jitter=0.3, weight_seed=0, resolution=128, grid_boundary=[-1.1, 1.1], stride=4)


def moments(a):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum()])


def main():
    plots, captured = ref_loader.load_plots()
    conf = config.mlp_config(SPEC['width'], 8)
    state = synth.make_state(conf, seed=SPEC['weight_seed'], jitter=SPEC['jitter'])
    model = ref_loader.build_model(conf, state, training=False)
    sdf = lambda x: model.implicit_network(x)[:, 0]            # evaluation/eval.py:76
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        plots.get_surface_sliding(path='', epoch=0, sdf=sdf, resolution=SPEC['resolution'],
                                  grid_boundary=SPEC['grid_boundary'], return_mesh=True, level=0)
    assert len(captured) == 1
    vol = captured[0]['volume'].astype(np.float32)
    n = vol.shape[0]
    assert vol.shape == (n, n, n) and n == 128
    k = SPEC['stride']
    fine_thr = 2 * (SPEC['grid_boundary'][1] - SPEC['grid_boundary'][0]) / n * 8 / 8     # threshold of the last level
    np.savez_compressed(OUT, spec=np.frombuffer(repr(sorted(SPEC.items())).encode(), dtype=np.uint8),
                        sub=vol[::k, ::k, ::k].copy(), moments=moments(vol),
                        near_moments=moments(vol[np.abs(vol) < fine_thr]),
                        near_count=np.asarray(int((np.abs(vol) < fine_thr).sum())),
                        spacing=np.asarray(captured[0]['spacing']),
                        # one full plane through the surface, so a wrong refinement mask cannot hide between samples
                        plane=vol[:, :, n // 2].copy())
    print('volume fixture: %s %.1f KB' % (vol.shape, os.path.getsize(OUT) / 1024))


if __name__ == '__main__':
    main()
