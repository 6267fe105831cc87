// Ray generation for image-mode inputs: pixel (u, v) + intrinsics + camera-to-world pose -> world-space unit
// direction, camera-space unit direction (its z is the depth scale) and the camera centre, one thread per ray.
//
// Replaces rend_util.get_camera_params + lift called twice per chunk -- once with the pose, once with the
// identity (reference: code/utils/rend_util.py:63-91,105-118; code/model/network.py:505-516) -- i.e. two
// bmm + normalise passes and ~20 element-wise launches.  4x4 poses only (the quaternion branch raises in Python).
#include "common.h"

// lift() with z = 1 and both normalisations, in the reference's operation order (shared by the two kernels below)
struct RayOut {
  float dir[3], dir_cam[3], cam[3];
};
__device__ __forceinline__ RayOut lift_ray(const float x, const float y, const float* __restrict__ K,
                                           const float* __restrict__ pose) {
  RayOut o;
  const float fx = K[0], sk = K[1], cx = K[2], fy = K[5], cy = K[6];
  const float xl = (x - cx + cy * sk / fy - sk * y / fy) / fx;
  const float yl = (y - cy) / fy;
  // camera frame (identity pose): normalise (xl, yl, 1); F.normalize clamps the norm at 1e-12
  {
    const float inv = 1.0f / fmaxf(sqrtf(xl * xl + yl * yl + 1.0f), 1e-12f);
    o.dir_cam[0] = xl * inv;
    o.dir_cam[1] = yl * inv;
    o.dir_cam[2] = inv;
  }
  // world frame: (pose * [xl, yl, 1, 1])[:3] - pose[:3, 3]  (bmm, then the subtraction, as the reference does)
  float w[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float t = pose[4 * r + 3];
    const float full = pose[4 * r + 0] * xl + pose[4 * r + 1] * yl + pose[4 * r + 2] + t;
    w[r] = full - t;
    o.cam[r] = t;
  }
  const float inv = 1.0f / fmaxf(sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]), 1e-12f);
  o.dir[0] = w[0] * inv;
  o.dir[1] = w[1] * inv;
  o.dir[2] = w[2] * inv;
  return o;
}

// One training batch of the pixel mode, assembled on the device: ray r of the batch is pixel (idx[r] mod HW) of the
// frame at position idx[r] / HW of the frame list.  Replaces the reference's per-pixel tables (SceneDatasetDN.
// convert_to_pixels, datasets/scene_dataset.py:269-307: ray_dirs, ray_dirs_tmp, ray_cam_loc and a 4x4 pose PER PIXEL of
// every image, built once on the CPU) and the DataLoader's gather + collate + host-to-device copy per step
// (__getitem__ 374-401): nothing per pixel is stored -- the rays of the sampled pixels are formed here from the per-frame
// pose / intrinsics (the same arithmetic, so the same values as the table's rows), and the ground-truth rows are
// gathered from the images, which live in HBM.
struct PixelRaysArgs {
  const long long* idx;      // [n] ray indices in [0, n_frames * HW)
  const int* frame_list;     // [n_frames] index into pose / intrinsics / images, or NULL (identity)
  const float* pose;         // [N,4,4]
  const float* K;            // [N,4,4]
  int n, width, hw, n_frames;
  float* dirs;               // [n,3]
  float* dirs_cam;           // [n,3]
  float* cam_loc;            // [n,3]
  float* ray_pose;           // [n,16]
  int* frame_pos;            // [n]  idx / HW (what the reference's ray_frame_idx holds)
  const float* src[4];       // ground-truth images [n_frames', HW, C_k] (rows in the order of the frame list) or NULL
  float* dst[4];             // [n, C_k]
  int ch[4];
};

__global__ void __launch_bounds__(256) msdf_pixel_rays_k(const PixelRaysArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const long long id = a.idx[i];
  if (id < 0 || id >= (long long)a.n_frames * a.hw) {
    // an index outside the table: no memory is touched for it; the ray comes back as zeros with frame position -1
#pragma unroll
    for (int r = 0; r < 3; ++r) a.dirs[3 * i + r] = a.dirs_cam[3 * i + r] = a.cam_loc[3 * i + r] = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) a.ray_pose[16 * (size_t)i + k] = 0.f;
    a.frame_pos[i] = -1;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (a.src[t] != nullptr)
        for (int c = 0; c < a.ch[t]; ++c) a.dst[t][(size_t)i * a.ch[t] + c] = 0.f;
    return;
  }
  const int f = (int)(id / a.hw), pix = (int)(id - (long long)f * a.hw);
  const int frame = a.frame_list ? a.frame_list[f] : f;
  // the reference's uv grid (scene_dataset.py:258-260): pixel p of a row-major image -> (u, v) = (column, row)
  const float x = (float)(pix % a.width), y = (float)(pix / a.width);
  const float* pose = a.pose + 16 * (size_t)frame;
  const RayOut o = lift_ray(x, y, a.K + 16 * (size_t)frame, pose);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    a.dirs[3 * i + r] = o.dir[r];
    a.dirs_cam[3 * i + r] = o.dir_cam[r];
    a.cam_loc[3 * i + r] = o.cam[r];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) a.ray_pose[16 * (size_t)i + k] = pose[k];
  a.frame_pos[i] = f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (a.src[t] != nullptr) {
      for (int c = 0; c < a.ch[t]; ++c) a.dst[t][(size_t)i * a.ch[t] + c] = a.src[t][(size_t)id * a.ch[t] + c];
    }
  }
}

__global__ void __launch_bounds__(256)
msdf_camera_rays_k(const float* __restrict__ uv, const float* __restrict__ pose, const float* __restrict__ K,
                   const int n, float* __restrict__ dirs, float* __restrict__ dirs_cam, float* __restrict__ cam_loc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const RayOut o = lift_ray(uv[2 * i], uv[2 * i + 1], K, pose);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    dirs[3 * i + r] = o.dir[r];
    dirs_cam[3 * i + r] = o.dir_cam[r];
    cam_loc[3 * i + r] = o.cam[r];
  }
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

extern "C" int msdf_pixel_rays(const int64_t* ray_idx, int n, const int32_t* frame_list, int n_frames,
                               const float* pose_all, const float* intrinsics_all, int width, int hw, float* ray_dirs,
                               float* ray_dirs_cam, float* cam_loc, float* ray_pose, int32_t* frame_pos,
                               const float* const* gt_src, float* const* gt_dst, const int32_t* gt_channels, int n_gt,
                               void* stream) {
  if (n < 0 || n_frames < 0 || width <= 0 || hw <= 0 || (hw % width) != 0 || n_gt < 0 || n_gt > 4) return MSDF_ERR_ARG;
  if (n == 0) return MSDF_OK;
  if (ray_idx == nullptr || pose_all == nullptr || intrinsics_all == nullptr || ray_dirs == nullptr ||
      ray_dirs_cam == nullptr || cam_loc == nullptr || ray_pose == nullptr || frame_pos == nullptr)
    return MSDF_ERR_ARG;
  PixelRaysArgs a;
  a.idx = (const long long*)ray_idx; a.frame_list = frame_list; a.pose = pose_all; a.K = intrinsics_all;
  a.n = n; a.width = width; a.hw = hw; a.n_frames = n_frames;
  a.dirs = ray_dirs; a.dirs_cam = ray_dirs_cam; a.cam_loc = cam_loc; a.ray_pose = ray_pose; a.frame_pos = frame_pos;
  for (int t = 0; t < 4; ++t) {
    a.src[t] = nullptr; a.dst[t] = nullptr; a.ch[t] = 0;
    if (t < n_gt && gt_src != nullptr && gt_src[t] != nullptr) {
      if (gt_dst == nullptr || gt_dst[t] == nullptr || gt_channels == nullptr || gt_channels[t] <= 0) return MSDF_ERR_ARG;
      a.src[t] = gt_src[t]; a.dst[t] = gt_dst[t]; a.ch[t] = gt_channels[t];
    }
  }
  msdf_pixel_rays_k<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(a);
  return msdf_check_launch();
}
