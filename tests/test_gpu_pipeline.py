"""The rows of SURVEY 8(f) composed the way the reference's runner composes them, on the GPU only: frames in HBM ->
pixel-mode batches (PixelRayTable, f-1) -> MonoSDFNetwork.forward -> fused MonoSDFLoss (f-2) -> Adam, then an
image-mode render of a held-out view (render_image, f-4) and the coarse-to-fine SDF volume (sdf_volume, f-3).
Synthetic scene (no dataset in the container): a sphere seen from inside, targets in closed form (bench.scene_targets)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _frames(n, side, seed):
    rng = np.random.default_rng(seed)
    poses, intr = [], []
    for _ in range(n):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        p = np.eye(4, dtype=np.float32)
        p[:3, :3], p[:3, 3] = q, rng.uniform(-0.15, 0.15, 3)
        k = np.eye(4, dtype=np.float32)
        k[0, 0] = k[1, 1] = 0.9 * side
        k[0, 2] = k[1, 2] = side / 2
        poses.append(torch.from_numpy(p))
        intr.append(torch.from_numpy(k))
    return torch.stack(poses), torch.stack(intr)


def test_training_render_and_volume_compose():
    import bench
    from monosdf_amd.conf import ConfigTree
    from monosdf_amd.model.loss import MonoSDFLoss
    from monosdf_amd.model.network import MonoSDFNetwork
    from monosdf_amd.utils import render
    from monosdf_amd.utils.ray_table import PixelRayTable
    from oracle import config
    side, n_frames = 24, 7
    poses, intr = _frames(n_frames, side, seed=3)
    # ground-truth images of every frame from the closed-form scene, evaluated on the table's own rays
    bare = PixelRayTable(poses, intr, (side, side))
    _, rays, _ = bare.batch(torch.arange(len(bare)).cuda())
    gt = bench.scene_targets(rays)                                      # [1, N*HW, C]
    imgs = {k: gt[k][0].reshape(n_frames, side * side, -1) for k in ('rgb', 'depth', 'mask', 'normal')}
    train_frames = list(range(n_frames - 1))
    table = PixelRayTable(poses, intr, (side, side), train_frames, **imgs)
    assert len(table) == (n_frames - 1) * side * side
    torch.manual_seed(0)
    model = MonoSDFNetwork(ConfigTree.from_dict(config.mlp_config(64, 8))).cuda().train()
    loss_fn = MonoSDFLoss('torch.nn.L1Loss', eikonal_weight=0.05, smooth_weight=0.005, depth_weight=0.1,
                          normal_l1_weight=0.05, normal_cos_weight=0.05)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)

    def held_out_psnr():
        f = n_frames - 1
        ys, xs = torch.meshgrid(torch.arange(side), torch.arange(side), indexing='ij')
        uv = torch.stack([xs.flatten(), ys.flatten()], -1)[None].float().cuda()
        inputs = {'uv': uv, 'pose': poses[f:f + 1].cuda(), 'intrinsics': intr[f:f + 1].cuda()}
        img = render.render_image(model, inputs, torch.zeros(1, dtype=torch.long, device='cuda'), side * side,
                                  split_n_pixels=200)
        mse = torch.mean((img['rgb_values'] - imgs['rgb'][f]) ** 2)
        return (-10.0 * torch.log(mse) / math.log(10.0)).item()

    before = held_out_psnr()
    losses = []
    gen = torch.Generator(device='cuda').manual_seed(1)
    for it in range(120):
        idx = torch.randint(len(table), (256,), device='cuda', generator=gen)
        indices, model_input, ground_truth = table.batch(idx)
        out = model(model_input, indices.long(), if_pixel_input=True)
        loss = loss_fn(out, {k: v[None] for k, v in ground_truth.items()}, if_pixel_input=True)['loss']
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    after = held_out_psnr()
    assert np.isfinite(losses).all()
    assert np.mean(losses[-20:]) < 0.5 * np.mean(losses[:5]), (losses[:5], losses[-5:])
    assert after > before + 3.0, (before, after)            # the held-out view got better: the pieces train something
    # the trained SDF's volume: finite, negative somewhere, positive somewhere (a surface exists inside the block)
    model.eval()
    with torch.no_grad():
        fn = lambda p: model.implicit_network(p)[:, 0]
        (origin, spacing, vol), = list(render.sdf_volume(fn, resolution=128, grid_boundary=(-1.1, 1.1), shard=False))
    assert vol.shape == (128, 128, 128) and np.isfinite(vol).all() and vol.min() < 0 < vol.max()
