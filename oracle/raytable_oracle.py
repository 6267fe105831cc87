"""TEST INFRASTRUCTURE (CPU): restatement of the reference dataset's per-pixel ray tables and the pixel branch of its
__getitem__ (datasets/scene_dataset.py:257-260 uv grid, 269-307 convert_to_pixels, 374-401 __getitem__).  Only
tests/ may import it; pinned by tests/golden/raytable_small.npz, which oracle/make_golden_raytable.py records from the
REAL reference methods."""
import numpy as np
import torch

from . import monosdf_oracle as mo


def uv_grid(img_res):
    """scene_dataset.py:258-260: (u, v) = (column, row) of a row-major image."""
    uv = np.mgrid[0:img_res[0], 0:img_res[1]].astype(np.int32)
    uv = torch.from_numpy(np.flip(uv, axis=0).copy()).float()
    return uv.reshape(2, -1).transpose(1, 0)


def pixel_table(pose_all, intrinsics_all, img_res, frame_idx_list):
    """convert_to_pixels (269-307): the per-pixel tables of the split's frames."""
    n = pose_all.shape[0]
    uv = uv_grid(img_res).unsqueeze(0).expand(n, -1, -1)
    hw = uv.shape[1]
    dirs, cam = mo.camera_rays(uv, pose_all, intrinsics_all)
    dirs_tmp, _ = mo.camera_rays(uv, torch.eye(4)[None].expand(n, -1, -1), intrinsics_all)
    sel = list(frame_idx_list)
    return {'ray_dirs': dirs[sel].reshape(-1, 3), 'ray_dirs_tmp': dirs_tmp[sel].reshape(-1, 3),
            'ray_cam_loc': cam.unsqueeze(1).expand(-1, hw, -1)[sel].reshape(-1, 3),
            'ray_pose': pose_all.unsqueeze(1).expand(-1, hw, -1, -1)[sel].reshape(-1, 4, 4),
            'ray_frame_idx': torch.from_numpy(np.repeat(np.arange(n, dtype=np.int32).reshape(-1, 1), hw, 1).flatten())}


def batch(table, gt_images, frame_idx_list, idx):
    """__getitem__ (374-401) + collate for ray indices idx: (indices, model_input, ground_truth)."""
    sel = list(frame_idx_list)
    sample = {k: table[k][idx] for k in ('ray_dirs', 'ray_dirs_tmp', 'ray_cam_loc', 'ray_pose')}
    gt = {k: v[sel].reshape(-1, v.shape[-1])[idx] for k, v in gt_images.items()}
    return table['ray_frame_idx'][idx], sample, gt
