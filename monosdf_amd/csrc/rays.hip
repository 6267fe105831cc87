// Ray generation for image-mode inputs: pixel (u, v) + intrinsics + camera-to-world pose -> world-space unit
// direction, camera-space unit direction (its z is the depth scale) and the camera centre, one thread per ray.
//
// Replaces rend_util.get_camera_params + lift called twice per chunk -- once with the pose, once with the
// identity (reference: code/utils/rend_util.py:63-91,105-118; code/model/network.py:505-516) -- i.e. two
// bmm + normalise passes and ~20 element-wise launches.  4x4 poses only (the quaternion branch raises in Python).
#include "common.h"

__global__ void __launch_bounds__(256)
msdf_camera_rays_k(const float* __restrict__ uv, const float* __restrict__ pose, const float* __restrict__ K,
                   const int n, float* __restrict__ dirs, float* __restrict__ dirs_cam, float* __restrict__ cam_loc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float fx = K[0], sk = K[1], cx = K[2], fy = K[5], cy = K[6];
  const float x = uv[2 * i], y = uv[2 * i + 1];
  // lift() with z = 1, in the reference's operation order
  const float xl = (x - cx + cy * sk / fy - sk * y / fy) / fx;
  const float yl = (y - cy) / fy;
  // camera frame (identity pose): normalise (xl, yl, 1); F.normalize clamps the norm at 1e-12
  {
    const float inv = 1.0f / fmaxf(sqrtf(xl * xl + yl * yl + 1.0f), 1e-12f);
    dirs_cam[3 * i + 0] = xl * inv;
    dirs_cam[3 * i + 1] = yl * inv;
    dirs_cam[3 * i + 2] = inv;
  }
  // world frame: (pose * [xl, yl, 1, 1])[:3] - pose[:3, 3]  (bmm, then the subtraction, as the reference does)
  float w[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float t = pose[4 * r + 3];
    const float full = pose[4 * r + 0] * xl + pose[4 * r + 1] * yl + pose[4 * r + 2] + t;
    w[r] = full - t;
    cam_loc[3 * i + r] = t;
  }
  const float inv = 1.0f / fmaxf(sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]), 1e-12f);
  dirs[3 * i + 0] = w[0] * inv;
  dirs[3 * i + 1] = w[1] * inv;
  dirs[3 * i + 2] = w[2] * inv;
}

extern "C" int msdf_camera_rays(const float* uv, const float* pose, const float* intrinsics, int n, float* ray_dirs,
                                float* ray_dirs_cam, float* cam_loc, void* stream) {
  if (n < 0) return MSDF_ERR_ARG;
  if (n == 0) return MSDF_OK;
  if (uv == nullptr || pose == nullptr || intrinsics == nullptr) return MSDF_ERR_ARG;
  msdf_camera_rays_k<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(uv, pose, intrinsics, n, ray_dirs, ray_dirs_cam,
                                                                        cam_loc);
  return msdf_check_launch();
}
