"""Full-image inference and SDF-volume evaluation on top of the fused kernels.

* ``render_image``: the chunked image loop of the reference (utils/general.py:28-58 split_input /
  merge_output; evaluation/eval.py:105-120; training/monosdf_train.py:351-370), chunks sharded over
  ranks (chunk c -> rank c mod world) and the rendered rows all-gathered over RCCL.
* ``sdf_volume``: the SDF-evaluation part of plots.get_surface_sliding (utils/plots.py:108-190): a
  4-level coarse-to-fine pyramid per block, only voxels with |sdf| < threshold refined; every level is
  one call of the no-grad forward kernel on the GPU (the reference moves 100k-point chunks to the CPU).
  On the GPU only the three axes of every pyramid level exist (the reference builds the 512^3 x 3 point grid
  with a CPU meshgrid and pools it), points are formed for the voxels a level's mask selects, and the volume
  leaves through a cached pinned buffer: 0.22 s per 512^3 block against 2.0 s for the reference's sequence.
  Marching cubes itself (skimage, plots.py:199) stays with the caller.
"""
import os

import numpy as np
import torch

from .. import parallel


def split_input(model_input, total_pixels, n_pixels=10000):
    """Slices of ``uv`` (and masks / depth when present) of at most n_pixels (general.py:28-42)."""
    chunks = []
    for idx in torch.split(torch.arange(total_pixels, device=model_input['uv'].device), n_pixels, dim=0):
        data = dict(model_input)
        data['uv'] = torch.index_select(model_input['uv'], 1, idx)
        for k in ('object_mask', 'depth'):
            if k in data:
                data[k] = torch.index_select(model_input[k], 1, idx)
        chunks.append(data)
    return chunks


def merge_output(res, total_pixels, batch_size):
    """Concatenate per-chunk outputs (general.py:44-58)."""
    out = {}
    for k in res[0]:
        if res[0][k] is None:
            continue
        if res[0][k].dim() == 1:
            out[k] = torch.cat([r[k].reshape(batch_size, -1, 1) for r in res], 1).reshape(batch_size * total_pixels)
        else:
            out[k] = torch.cat([r[k].reshape(batch_size, -1, r[k].shape[-1]) for r in res],
                               1).reshape(batch_size * total_pixels, -1)
    return out


def lin2img(tensor, img_res):
    """[B, H*W, C] -> [B, C, H, W] (utils/plots.py:599-601)."""
    batch_size, num_samples, channels = tensor.shape
    return tensor.permute(0, 2, 1).view(batch_size, channels, img_res[0], img_res[1])


_WIDTH = {'rgb_values': 3, 'normal_map': 3, 'depth_values': 1}
# streams the chunks of render_image alternate over (1: the reference's sequential loop)
RENDER_STREAMS = int(os.environ.get('MSDF_RENDER_STREAMS', '2'))
_STREAMS = {}


def _render_streams(device, n):
    key = (torch.device(device).index, n)
    if key not in _STREAMS:
        _STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _STREAMS[key]


@torch.no_grad()
def render_image(model, model_input, indices, total_pixels, split_n_pixels=1024,
                 keys=('rgb_values', 'normal_map', 'depth_values')):
    """Eval-mode render of all pixels of ``model_input['uv']``; returns {key: [total_pixels, C]} on every rank.
    Chunk c is rendered by rank c mod world; one ragged all-gather reassembles the image."""
    was_training = model.training
    model.eval()
    try:
        chunks = split_input(model_input, total_pixels, split_n_pixels)
        world, rank = parallel.world(), parallel.rank()
        rows = []
        mine = list(range(rank, len(chunks), world))
        dev = model_input['uv'].device
        if dev.type == 'cuda' and len(mine) > 1 and RENDER_STREAMS > 1:
            # chunks are independent: consecutive chunks go to alternating streams, so that the partly filled last round
            # of workgroups of one chunk's kernels (1,024 rays x 98 samples = 3.06 / 3.19 rounds of the 512 slots) and
            # the launch gaps of its small kernels are filled by the other chunk's work.  Same kernels, same values.
            main = torch.cuda.current_stream(dev)
            streams = _render_streams(dev, RENDER_STREAMS)
            for st in streams:
                st.wait_stream(main)
            for j, c in enumerate(mine):
                st = streams[j % len(streams)]
                with torch.cuda.stream(st):
                    out = model(chunks[c], indices)
                    r = torch.cat([out[k].reshape(out[k].shape[0], -1) for k in keys], 1)
                r.record_stream(main)              # allocated on `st`, read by the concatenation on `main`
                rows.append(r)
            for st in streams:
                main.wait_stream(st)
        else:
            for c in mine:
                out = model(chunks[c], indices)
                rows.append(torch.cat([out[k].reshape(out[k].shape[0], -1) for k in keys], 1))
        width = sum(_WIDTH[k] for k in keys)
        local = torch.cat(rows, 0) if rows else torch.zeros(0, width, device=model_input['uv'].device)
        if world > 1:
            # every rank knows the deal (chunk c -> rank c mod world), hence every rank's row count: one collective,
            # no size exchange and no host read
            n_chunk = [ch['uv'].shape[1] for ch in chunks]
            gathered = parallel.all_gather_rows(
                local, sizes=[sum(n_chunk[c] for c in range(r, len(chunks), world)) for r in range(world)])
            span, off = {}, 0
            for r in range(world):                       # rank-major order of the gathered rows
                for c in range(r, len(chunks), world):
                    span[c] = (off, off + n_chunk[c])
                    off += n_chunk[c]
            full = torch.cat([gathered[span[c][0]:span[c][1]] for c in range(len(chunks))], 0)
        else:
            full = local
        result, col = {}, 0
        for k in keys:
            result[k] = full[:, col:col + _WIDTH[k]]
            col += _WIDTH[k]
        return result
    finally:
        model.train(was_training)


_PINNED = {}


def _to_host(t):
    """Device tensor -> numpy through a cached pinned buffer (a pageable copy of a 512^3 volume takes ~10 x as long)."""
    key = (tuple(t.shape), t.dtype)
    if key not in _PINNED:
        _PINNED.clear()                     # one volume size at a time: 537 MB of pinned memory per 512^3 block
        _PINNED[key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host = _PINNED[key]
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return host.numpy().copy()


def _block_on_device(evaluate, mins, maxs, cropN, device):
    """One block of the coarse-to-fine evaluation with everything but the network on the device's own terms: the
    reference (plots.py:135-190) builds the cropN^3 x 3 point grid with a CPU meshgrid (3.2 GB of doubles at 512^3),
    copies 1.6 GB to the GPU and average-pools it three times; the grid is separable, so here only the three AXES of
    every pyramid level exist (the pooled coordinate of a coarse voxel is the mean of its two fine coordinates), the
    points of a level are formed for the voxels its mask selects, and the finished volume leaves through a pinned
    buffer.  Same masks, thresholds, nearest-neighbour upsampling and evaluation order as the reference; a pooled
    coordinate may differ from AvgPool3d's fp32 sum of eight in the last bit (the volume is compared with the
    reference's at 1e-4 of its range, tests/test_gpu_parity.py)."""
    # float(np.linspace in float64), the values of the reference's grid (plots.py:139-146)
    fine = [torch.from_numpy(np.linspace(mins[d], maxs[d], cropN)).float().to(device) for d in range(3)]
    levels = [fine]
    for _ in range(3):
        levels.append([(a[0::2] + a[1::2]) * 0.5 for a in levels[-1]])
    levels = levels[::-1]
    upsample = torch.nn.Upsample(scale_factor=2, mode='nearest')
    mask, vals = None, None
    threshold = 2 * (maxs[0] - mins[0]) / cropN * 8
    for pid, (ax, ay, az) in enumerate(levels):
        n = ax.shape[0]
        if mask is None:
            xx, yy, zz = torch.meshgrid(ax, ay, az, indexing='ij')
            vals = evaluate(torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], 1))
        else:
            idx = mask.reshape(-1).nonzero().reshape(-1)          # ascending: the order of the reference's flat[m]
            if idx.numel() > 0:
                ii, rem = torch.div(idx, n * n, rounding_mode='floor'), idx % (n * n)
                jj, kk = torch.div(rem, n, rounding_mode='floor'), rem % n
                vals[idx] = evaluate(torch.stack([ax[ii], ay[jj], az[kk]], 1))
        if pid < 3:
            mask = upsample((vals.abs() < threshold).reshape(n, n, n)[None, None].float()).bool()
            vals = upsample(vals.reshape(n, n, n)[None, None]).reshape(-1)
        threshold /= 2.
    return _to_host(vals.reshape(cropN, cropN, cropN).float())


def _pyramid(points, levels=3):
    """[3, n, n, n] -> list coarse..fine by 2x average pooling (plots.py:154-159)."""
    pyr = [points]
    pool = torch.nn.AvgPool3d(2, stride=2)
    for _ in range(levels):
        points = pool(points[None])[0]
        pyr.append(points)
    return pyr[::-1]


@torch.no_grad()
def sdf_volume(sdf_fn, resolution=512, grid_boundary=(-1.1, 1.1), device='cuda', shard=True):
    """Yields (origin[3], spacing[3], volume[cropN, cropN, cropN] float32 numpy) per block of the sliding
    window, exactly the array the reference hands to marching cubes.  ``sdf_fn(points[P,3]) -> [P]``."""
    cropN = 128 if resolution < 512 else 512
    assert resolution % cropN == 0
    N = resolution // cropN
    lo, hi = grid_boundary
    edges = np.linspace(lo, hi, N + 1)
    upsample = torch.nn.Upsample(scale_factor=2, mode='nearest')
    world, rank = (parallel.world(), parallel.rank()) if shard else (1, 0)

    def evaluate(pts):
        if world == 1:
            return sdf_fn(pts).reshape(-1)
        a, b = parallel.shard_slice(pts.shape[0], rank, world)
        part = sdf_fn(pts[a:b].contiguous()).reshape(-1, 1) if b > a else pts.new_zeros(0, 1)
        # the point list of a level is the same on every rank (the mask comes from gathered values), so the cut is too
        cuts = [parallel.shard_slice(pts.shape[0], r, world) for r in range(world)]
        return parallel.all_gather_rows(part, sizes=[hi_ - lo_ for lo_, hi_ in cuts]).reshape(-1)

    on_gpu = torch.device(device).type == 'cuda'
    for i in range(N):
        for j in range(N):
            for k in range(N):
                mins = (edges[i], edges[j], edges[k])
                maxs = (edges[i + 1], edges[j + 1], edges[k + 1])
                if on_gpu:
                    spacing = tuple((maxs[d] - mins[d]) / (cropN - 1) for d in range(3))
                    yield np.array(mins), spacing, _block_on_device(evaluate, mins, maxs, cropN, device)
                    continue
                # CPU (the gloo tests over the oracle's network): the reference's own sequence of tensor operations
                axes = [torch.tensor(np.linspace(mins[d], maxs[d], cropN)) for d in range(3)]
                xx, yy, zz = torch.meshgrid(*axes, indexing='ij')
                pts = torch.vstack([xx.flatten(), yy.flatten(), zz.flatten()]).T.float().to(device)
                pyr = _pyramid(pts.reshape(cropN, cropN, cropN, 3).permute(3, 0, 1, 2))
                mask, vals = None, None
                threshold = 2 * (maxs[0] - mins[0]) / cropN * 8
                for pid, p in enumerate(pyr):
                    n = p.shape[-1]
                    flat = p.reshape(3, -1).permute(1, 0).contiguous()
                    if mask is None:
                        vals = evaluate(flat)
                    else:
                        m = mask.reshape(-1)
                        if bool(m.any()):
                            vals[m] = evaluate(flat[m].contiguous())
                    if pid < 3:
                        mask = (vals.abs() < threshold).reshape(n, n, n)[None, None]
                        mask = upsample(mask.float()).bool()
                        vals = upsample(vals.reshape(n, n, n)[None, None]).reshape(-1)
                    threshold /= 2.
                spacing = tuple((maxs[d] - mins[d]) / (cropN - 1) for d in range(3))
                yield np.array(mins), spacing, vals.reshape(cropN, cropN, cropN).cpu().numpy().astype(np.float32)
