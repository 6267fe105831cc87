"""Ray generation for image-mode inputs (reference: code/utils/rend_util.py:63-91, 105-118).
The step immediately before the hot path; a handful of [N]-sized tensor ops."""
import torch
from torch.nn import functional as F


def lift(x, y, z, intrinsics):
    fx, fy = intrinsics[:, 0, 0].unsqueeze(-1), intrinsics[:, 1, 1].unsqueeze(-1)
    cx, cy = intrinsics[:, 0, 2].unsqueeze(-1), intrinsics[:, 1, 2].unsqueeze(-1)
    sk = intrinsics[:, 0, 1].unsqueeze(-1)
    x_lift = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    y_lift = (y - cy) / fy * z
    return torch.stack((x_lift, y_lift, z, torch.ones_like(z)), dim=-1)


def get_camera_params(uv, pose, intrinsics):
    if pose.shape[1] == 7:
        raise NotImplementedError('quaternion poses are not used on this path')
    cam_loc = pose[:, :3, 3]
    batch_size, num_samples, _ = uv.shape
    x_cam = uv[:, :, 0].view(batch_size, -1)
    y_cam = uv[:, :, 1].view(batch_size, -1)
    z_cam = torch.ones_like(x_cam)
    pts = lift(x_cam, y_cam, z_cam, intrinsics).permute(0, 2, 1)
    world = torch.bmm(pose, pts).permute(0, 2, 1)[:, :, :3]
    ray_dirs = F.normalize(world - cam_loc[:, None, :], dim=2)
    return ray_dirs, cam_loc


def get_psnr(img1, img2, normalize_rgb=False):
    if normalize_rgb:
        img1, img2 = (img1 + 1.) / 2., (img2 + 1.) / 2.
    mse = torch.mean((img1 - img2) ** 2)
    return -10. * torch.log(mse) / torch.log(torch.tensor(10., device=mse.device))
