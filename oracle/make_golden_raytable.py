"""Fixture of the dataset's per-pixel ray tables recorded from the REAL reference (container-only, TEST INFRASTRUCTURE):

    python -m oracle.make_golden_raytable

Runs the reference's own `SceneDatasetDN.convert_to_pixels`, `__getitem__` (pixel branch) and `collate_fn`
(datasets/scene_dataset.py:269-307, 374-401, 438-...) -- as unbound methods on an object that carries the attributes
__init__ would have read from image files (poses, intrinsics, images: synthetic here, from numpy seeds) -- and stores
inputs, sampled ray indices and the collated batch.  The uv grid is built by the three lines of __init__ that build it
(257-260), restated in oracle/raytable_oracle.uv_grid."""
import contextlib
import io
import os
import types

import numpy as np
import torch

from . import raytable_oracle as ro, ref_loader

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'raytable_small.npz')
SPEC = dict(n_images=5, img_res=(12, 16), frames=(0, 2, 3), seed=21, n_batch=97)


def synth(spec):
    rng = np.random.default_rng(spec['seed'])
    n, (h, w) = spec['n_images'], spec['img_res']
    poses, intr = [], []
    for _ in range(n):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        p = np.eye(4)
        p[:3, :3], p[:3, 3] = q, rng.uniform(-0.5, 0.5, 3)
        k = np.eye(4)
        k[0, 0], k[1, 1] = rng.uniform(10, 20), rng.uniform(10, 20)
        k[0, 1] = rng.uniform(-0.2, 0.2)                    # skew
        k[0, 2], k[1, 2] = w / 2 + rng.uniform(-1, 1), h / 2 + rng.uniform(-1, 1)
        poses.append(torch.from_numpy(p).float())
        intr.append(torch.from_numpy(k).float())
    img = lambda c: [torch.from_numpy(rng.uniform(0, 1, (h * w, c))).float() for _ in range(n)]
    return poses, intr, {'rgb': img(3), 'depth': img(1), 'mask': img(1), 'normal': img(3)}


def main():
    ref_loader.load()
    import sys
    stub = types.ModuleType('termcolor')
    stub.colored = lambda s, *a, **k: s
    sys.modules.setdefault('termcolor', stub)
    if 'torchvision' not in sys.modules:                    # utils/general.py imports torchvision.transforms (unused here)
        tv = types.ModuleType('torchvision')
        tv.transforms = types.ModuleType('torchvision.transforms')
        tv.transforms.ToPILImage = lambda *a, **k: None
        sys.modules['torchvision'], sys.modules['torchvision.transforms'] = tv, tv.transforms
    # the reference's datasets/scene_dataset.py, loaded by path ("datasets" is also an installed package's name)
    import importlib.util
    path = os.path.join(ref_loader.REF_ROOT, 'datasets', 'scene_dataset.py')
    mod_spec = importlib.util.spec_from_file_location('ref_scene_dataset', path)
    mod = importlib.util.module_from_spec(mod_spec)
    with contextlib.redirect_stdout(io.StringIO()):
        mod_spec.loader.exec_module(mod)
    SceneDatasetDN = mod.SceneDatasetDN
    spec = SPEC
    poses, intr, imgs = synth(spec)
    ds = types.SimpleNamespace()
    ds.img_res, ds.pose_all, ds.intrinsics_all = list(spec['img_res']), poses, intr
    ds.frame_idx_list = list(spec['frames'])
    ds.uv = ro.uv_grid(spec['img_res'])
    ds.rgb_images, ds.depth_images, ds.mask_images, ds.normal_images = (imgs[k] for k in ('rgb', 'depth', 'mask', 'normal'))
    SceneDatasetDN.convert_to_pixels(ds)
    ds.if_pixel, ds.num_views = True, -1
    rng = np.random.default_rng(spec['seed'] + 1)
    idx = rng.integers(0, ds.total_pixels, spec['n_batch'])
    items = [SceneDatasetDN.__getitem__(ds, int(i)) for i in idx]
    indices, sample, gt = SceneDatasetDN.collate_fn(ds, items, if_pixel=True)
    rec = {'spec': np.frombuffer(repr(sorted(spec.items())).encode(), dtype=np.uint8), 'idx': idx.astype(np.int64),
           'total_pixels': np.asarray(ds.total_pixels), 'indices': np.asarray(indices)}
    rec.update({'in.pose': torch.stack(poses).numpy(), 'in.intrinsics': torch.stack(intr).numpy()})
    rec.update({'in.' + k: torch.stack(v).numpy() for k, v in imgs.items()})
    rec.update({'sample.' + k: v.numpy() for k, v in sample.items()})
    rec.update({'gt.' + k: v.numpy() for k, v in gt.items()})
    np.savez_compressed(OUT, **rec)
    print('raytable_small %.1f KB; batch keys %s / %s' % (os.path.getsize(OUT) / 1024, sorted(sample), sorted(gt)))


if __name__ == '__main__':
    main()
