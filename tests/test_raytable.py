"""SURVEY 8(f)-1, second half: the dataset's per-pixel ray tables and pixel-mode batches.

tests/golden/raytable_small.npz was recorded from the REAL reference methods (SceneDatasetDN.convert_to_pixels,
__getitem__, collate_fn; oracle/make_golden_raytable.py).  CPU: the oracle restatement reproduces it bit for bit.
GPU: `monosdf_amd.utils.ray_table.PixelRayTable.batch` (one HIP launch, nothing stored per pixel) returns the same batch."""
import ast
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, TOL, rel_err
from oracle import raytable_oracle as ro


def _fixture():
    z = np.load(os.path.join(GOLDEN, 'raytable_small.npz'))
    spec = dict(ast.literal_eval(bytes(z['spec']).decode()))
    t = lambda k: torch.from_numpy(z[k])
    return z, spec, t


def test_oracle_reproduces_the_reference_tables():
    z, spec, t = _fixture()
    table = ro.pixel_table(t('in.pose'), t('in.intrinsics'), spec['img_res'], spec['frames'])
    assert table['ray_dirs'].shape[0] == int(z['total_pixels']) == len(spec['frames']) * spec['img_res'][0] * spec['img_res'][1]
    imgs = {k: t('in.' + k) for k in ('rgb', 'depth', 'mask', 'normal')}
    indices, sample, gt = ro.batch(table, imgs, spec['frames'], t('idx'))
    assert np.array_equal(indices.numpy(), z['indices'])
    for k, v in sample.items():
        assert torch.equal(v, t('sample.' + k)), k
    for k, v in gt.items():
        assert torch.equal(v, t('gt.' + k)), k


@pytest.mark.gpu
def test_pixel_batches_on_the_device_match_the_reference(errlog):
    from monosdf_amd.utils.ray_table import PixelRayTable
    z, spec, t = _fixture()
    imgs = {k: t('in.' + k) for k in ('rgb', 'depth', 'mask', 'normal')}
    tab = PixelRayTable(t('in.pose'), t('in.intrinsics'), spec['img_res'], spec['frames'], **imgs)
    assert len(tab) == int(z['total_pixels'])
    indices, sample, gt = tab.batch(t('idx').cuda())
    assert np.array_equal(indices.cpu().numpy(), z['indices'])
    assert set(sample) == {'ray_dirs', 'ray_dirs_tmp', 'ray_cam_loc', 'ray_pose'} and set(gt) == set(imgs)
    for k, v in sample.items():
        ref = t('sample.' + k)
        assert v.shape == ref.shape, k
        err = rel_err(v, ref)
        errlog('raytable', 'raytable_small', k, err, 2e-6)
        assert err <= 2e-6, (k, err)                  # two fp32 roundings of the lift / normalise chain (as test_ray_generation)
    for k, v in gt.items():
        assert torch.equal(v.cpu(), t('gt.' + k)), k  # gathered rows: exact
    # every pixel of the split: the whole table the reference would have stored
    table = ro.pixel_table(t('in.pose'), t('in.intrinsics'), spec['img_res'], spec['frames'])
    all_idx = torch.arange(len(tab))
    indices, sample, gt = tab.batch(all_idx.cuda())
    assert torch.equal(indices.cpu().long(), table['ray_frame_idx'][:len(tab)].long())
    for k in sample:
        assert rel_err(sample[k], table[k]) <= 2e-6, k
    # no ground truth given, an empty batch
    bare = PixelRayTable(t('in.pose'), t('in.intrinsics'), spec['img_res'])
    ind, smp, g = bare.batch(torch.zeros(0, dtype=torch.long).cuda())
    assert ind.shape == (0,) and smp['ray_dirs'].shape == (0, 3) and smp['ray_pose'].shape == (0, 4, 4) and g == {}
    ind, smp, g = bare.batch(torch.tensor([0, len(bare) - 1]).cuda())
    assert ind.tolist() == [0, spec['n_images'] - 1] and g == {}
    # an index outside the table touches no memory: zeros and frame position -1
    ind, smp, g = tab.batch(torch.tensor([len(tab), -1, 5]).cuda())
    assert ind.tolist()[:2] == [-1, -1] and ind.tolist()[2] == 0
    assert float(smp['ray_dirs'][:2].abs().max()) == 0.0 and float(g['rgb'][:2].abs().max()) == 0.0
    assert float(smp['ray_dirs'][2].norm()) > 0.99


@pytest.mark.gpu
def test_pixel_batch_feeds_the_model():
    """A batch from the table goes straight into MonoSDFNetwork.forward(if_pixel_input=True) and the fused loss."""
    from monosdf_amd.conf import ConfigTree
    from monosdf_amd.model.loss import MonoSDFLoss
    from monosdf_amd.model.network import MonoSDFNetwork
    from monosdf_amd.utils.ray_table import PixelRayTable
    from oracle import config
    z, spec, t = _fixture()
    imgs = {k: t('in.' + k) for k in ('rgb', 'depth', 'mask', 'normal')}
    tab = PixelRayTable(t('in.pose'), t('in.intrinsics'), spec['img_res'], spec['frames'], **imgs)
    m = MonoSDFNetwork(ConfigTree.from_dict(config.mlp_config(64, 8))).cuda().train()
    indices, sample, gt = tab.batch(torch.randint(len(tab), (32,)).cuda())
    out = m(sample, indices.long(), if_pixel_input=True)
    loss = MonoSDFLoss('torch.nn.L1Loss', eikonal_weight=0.05)(out, {k: v[None] for k, v in gt.items()},
                                                               if_pixel_input=True)['loss']
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
