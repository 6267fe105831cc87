"""Pixel-mode training batches assembled on the GPU (second half).

The reference's dataset (datasets/scene_dataset.py:269-307, `convert_to_pixels`) builds, once and on the CPU, a table
with one row PER PIXEL of every training image -- ray direction, camera-frame direction, camera centre and a full 4x4
pose (112 bytes per pixel before the ground truth: 6.6 GB for 400 images of 384 x 384) -- and its DataLoader then
gathers a batch of rows (`__getitem__` 374-401), collates and copies it to the GPU every step.

Here nothing per pixel is stored: the per-frame poses / intrinsics and the ground-truth images live in HBM (288 GB: a
whole scene fits), and ONE launch (`msdf_pixel_rays`, csrc/rays.hip) forms the rays of the sampled pixels with the same
arithmetic as the table's rows and gathers their ground truth.  `batch()` returns what the reference's collate hands
the training loop: (indices, model_input, ground_truth) with the same keys and shapes."""
import ctypes as C

import torch

from .. import _lib


class PixelRayTable:
    GT_KEYS = (('rgb', 3), ('depth', 1), ('mask', 1), ('normal', 3))

    def __init__(self, pose_all, intrinsics_all, img_res, frame_idx_list=None, rgb=None, depth=None, mask=None,
                 normal=None, device='cuda'):
        """pose_all / intrinsics_all: [N,4,4] (or lists of [4,4]); img_res = (H, W); frame_idx_list: the frames of this
        split (default: all); rgb / depth / mask / normal: per-frame images [N,HW,C] (or lists), any of them None."""
        stack = lambda t: (torch.stack(list(t)) if isinstance(t, (list, tuple)) else t)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('monosdf_amd: PixelRayTable lives on the GPU (the HIP path has no CPU fallback)')
        f32 = dict(device=self.device, dtype=torch.float32)
        self.pose = stack(pose_all).to(**f32).contiguous()
        self.intrinsics = stack(intrinsics_all).to(**f32).contiguous()
        n_all = self.pose.shape[0]
        if self.pose.shape[1:] != (4, 4) or self.intrinsics.shape != (n_all, 4, 4):
            raise NotImplementedError('monosdf_amd: 4x4 pose / intrinsics matrices per frame')
        self.H, self.W = int(img_res[0]), int(img_res[1])
        self.hw = self.H * self.W
        frames = list(range(n_all)) if frame_idx_list is None else [int(f) for f in frame_idx_list]
        if any(f < 0 or f >= n_all for f in frames):
            raise IndexError('frame index outside the %d frames' % n_all)
        self.frame_list = torch.tensor(frames, device=self.device, dtype=torch.int32)
        self.n_frames = len(frames)
        self.total_pixels = self.n_frames * self.hw
        # ground truth of the split's frames, rows in frame-list order: [n_frames * HW, C]
        self.gt = {}
        sel = torch.tensor(frames, dtype=torch.long)
        for (key, ch), img in zip(self.GT_KEYS, (rgb, depth, mask, normal)):
            if img is None:
                continue
            img = stack(img)
            if img.shape[0] != n_all or img.shape[1] != self.hw or img.shape[2] != ch:
                raise ValueError('%s images must be [%d, %d, %d], got %s' % (key, n_all, self.hw, ch, tuple(img.shape)))
            self.gt[key] = img[sel.to(img.device)].reshape(-1, ch).to(**f32).contiguous()

    def __len__(self):
        return self.total_pixels

    def batch(self, idx):
        """idx: int64 ray indices in [0, total_pixels) -> (indices [B] int32 frame positions, model_input, ground_truth)."""
        idx = idx.to(device=self.device, dtype=torch.int64).contiguous().reshape(-1)
        n = idx.numel()
        f32 = dict(device=self.device, dtype=torch.float32)
        dirs = torch.empty(3, n, 3, **f32)
        pose = torch.empty(n, 4, 4, **f32)
        frame_pos = torch.empty(n, device=self.device, dtype=torch.int32)
        keys = [k for k, _ in self.GT_KEYS if k in self.gt]
        out = {k: torch.empty(n, self.gt[k].shape[1], **f32) for k in keys}
        P = C.c_void_p * 4
        src = P(*([self.gt[k].data_ptr() for k in keys] + [None] * (4 - len(keys))))
        dst = P(*([out[k].data_ptr() if n else None for k in keys] + [None] * (4 - len(keys))))
        chs = (C.c_int32 * 4)(*([self.gt[k].shape[1] for k in keys] + [0] * (4 - len(keys))))
        _lib.call('msdf_pixel_rays', _lib.ptr(idx), n, _lib.ptr(self.frame_list), self.n_frames, _lib.ptr(self.pose),
                  _lib.ptr(self.intrinsics), self.W, self.hw, _lib.ptr(dirs[0]), _lib.ptr(dirs[1]), _lib.ptr(dirs[2]),
                  _lib.ptr(pose), _lib.ptr(frame_pos), src, dst, chs, len(keys), _lib.stream_ptr())
        model_input = {'ray_dirs': dirs[0], 'ray_dirs_tmp': dirs[1], 'ray_cam_loc': dirs[2], 'ray_pose': pose}
        return frame_pos, model_input, out
