"""Image-side helpers of the reference's utils/rend_util.py that the render drivers use.
Ray generation (get_camera_params + lift, rend_util.py:63-91,105-118) is the HIP kernel `msdf_camera_rays`
(csrc/rays.hip, ops.camera_rays)."""
import torch


def get_camera_params(uv, pose, intrinsics):
    """Reference signature (rend_util.py:63): uv [1,n,2], pose [1,4,4], intrinsics [1,4,4] ->
    (ray_dirs [1,n,3], cam_loc [1,3])."""
    from .. import ops
    if pose.shape[0] != 1 or pose.shape[1:] != (4, 4):
        raise NotImplementedError('one 4x4 camera-to-world pose per call (quaternion poses are not used on this path)')
    dirs, _, cam = ops.camera_rays(uv[0], pose[0], intrinsics[0])
    return dirs.unsqueeze(0), cam[:1]


def get_psnr(img1, img2, normalize_rgb=False):
    if normalize_rgb:
        img1, img2 = (img1 + 1.) / 2., (img2 + 1.) / 2.
    mse = torch.mean((img1 - img2) ** 2)
    return -10. * torch.log(mse) / torch.log(torch.tensor(10., device=mse.device))
