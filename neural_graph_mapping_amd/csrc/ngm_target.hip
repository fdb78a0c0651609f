// Training-target sampler, device part (NeuralGraphMap._sample_target_mv, rm.py:1259-1459).
// The random draws (field subsets, sphere offsets, keyframe ids, pixel uniforms) stay with the caller's
// torch generator, exactly as in the reference; these two kernels replace the ~40 small tensor ops between them:
//   k_target_visibility : which keyframes see which field (20 points on the field sphere projected into every
//                         keyframe, depth test against the keyframe's depth image) + the 2-D bounding box of
//                         the projections per (field, keyframe)                                  rm.py:1321-1392
//   k_target_rays       : per sampled ray: pixel inside the box, near/far from the field sphere, RGB-D target,
//                         ray distance of the depth, masks, termination target                  rm.py:1394-1459
// HBM-bound gathers; one thread per (field, keyframe) resp. per ray.
#include "ngm_device.h"
#include "ngm_launch.h"

#pragma clang fp contract(off)

struct Cam3 { float x, y, z; };

// p_c = R^T (p_w - t)  (utils.transform_points(..., inv=True), utils.py:279-282), T row-major 4x4
__device__ __forceinline__ Cam3 world_to_cam(const float* T, float px, float py, float pz) {
  const float dx = px - T[3], dy = py - T[7], dz = pz - T[11];
  Cam3 c;
  c.x = T[0] * dx + T[4] * dy + T[8] * dz;
  c.y = T[1] * dx + T[5] * dy + T[9] * dz;
  c.z = T[2] * dx + T[6] * dy + T[10] * dz;
  return c;
}

__global__ void k_target_visibility(ngm_keyframes kf, int F, const float* __restrict__ field_pos, int num_offsets,
                                    const float* __restrict__ offsets, float radius, uint8_t* __restrict__ kf_mask,
                                    float* __restrict__ bbox) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= F * kf.num_frames) return;
  const int f = idx / kf.num_frames, c = idx - f * kf.num_frames;
  const float* T = kf.c2ws + 16 * (int64_t)c;
  const float* img = kf.rgbd + kf.frame_to_store[c] * (int64_t)kf.height * kf.width * 4;
  // Camera.project_points(points, "opengl") with the default pixel centre 0.5 (camera.py:119-154,176-180)
  const float cxp = kf.cx + 0.5f, cyp = kf.cy + 0.5f;
  const float px = field_pos[3 * f], py = field_pos[3 * f + 1], pz = field_pos[3 * f + 2];
  bool in_front = false, in_front_depth = false, in_frustum = false;
  float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
  for (int s = 0; s < num_offsets; ++s) {
    const float wx = px + offsets[3 * s] * radius * 1.0f, wy = py + offsets[3 * s + 1] * radius * 1.0f,
                wz = pz + offsets[3 * s + 2] * radius * 1.0f;
    const Cam3 p = world_to_cam(T, wx, wy, wz);
    const float depth = -p.z;
    const float xh = kf.fx * p.x + (-cxp) * p.z, yh = (-kf.fy) * p.y + (-cyp) * p.z, zh = -p.z;
    const float X = xh / zh, Y = yh / zh;
    const int xi = (int)X, yi = (int)Y;                            // .int(): truncation towards zero (rm.py:1337)
    const bool valid = xi >= 0 && xi < kf.width && yi >= 0 && yi < kf.height;
    const float kd = valid ? img[((int64_t)yi * kf.width + xi) * 4 + 3] : 0.f;
    in_front |= depth > 0.f;
    in_front_depth |= depth < kd;
    in_frustum |= valid;
    mnx = fminf(mnx, X); mny = fminf(mny, Y); mxx = fmaxf(mxx, X); mxy = fmaxf(mxy, Y);
  }
  kf_mask[idx] = (in_front && in_front_depth && in_frustum) ? 1 : 0;
  // boxes are stored clamped to the image as the reference does before gathering them (rm.py:1388-1392)
  reinterpret_cast<float4*>(bbox)[idx] = make_float4(fmaxf(mnx, 0.f), fmaxf(mny, 0.f), fminf(mxx, (float)kf.width),
                                                     fminf(mxy, (float)kf.height));
}

__global__ void k_target_rays(ngm_keyframes kf, int F, int R, const float* __restrict__ field_pos, float radius,
                              const float* __restrict__ bbox, const int64_t* __restrict__ frame_cids,
                              const float* __restrict__ u_xy, ngm_target_out o) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)F * R) return;
  const int f = (int)(idx / R);
  const int64_t c = frame_cids[idx];
  const float4 bb = reinterpret_cast<const float4*>(bbox)[(int64_t)f * kf.num_frames + c];
  const float ux = u_xy[2 * idx], uy = u_xy[2 * idx + 1];
  const float x = (bb.z - bb.x) * ux + bb.x, y = (bb.w - bb.y) * uy + bb.y;          // rm.py:1400-1402
  int j = min((int)x, kf.width - 1), i = min((int)y, kf.height - 1);                 // rm.py:1403-1407
  // (a negative index can only come from a keyframe that does not see the field; torch would wrap it around)
  const int jc = max(j, 0), ic = max(i, 0);
  const float* T = kf.c2ws + 16 * c;
  if (o.c2ws) {
    float4* dst = reinterpret_cast<float4*>(o.c2ws + 16 * idx);
    const float4* src = reinterpret_cast<const float4*>(T);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
  }
  o.ijs[2 * idx] = i; o.ijs[2 * idx + 1] = j;
  const Cam3 pc = world_to_cam(T, field_pos[3 * f], field_pos[3 * f + 1], field_pos[3 * f + 2]);
  // ijs_to_directions, OpenGL (camera.py:186-203) and the OpenCV z component for depth_to_distance (:339-340)
  const float dx = ((float)j - kf.cx) / kf.fx, dy = ((float)i - kf.cy) / kf.fy;
  const float nrm = fmaxf(sqrtf(dx * dx + dy * dy + 1.0f), 1e-12f);
  const float gx = dx / nrm, gy = (-dy) / nrm, gz = -1.0f / nrm;
  const float center = pc.x * gx + pc.y * gy + pc.z * gz;
  const float nearv = fmaxf(center - radius, 0.f), farv = fmaxf(center + radius, 0.f);
  const float4 px = reinterpret_cast<const float4*>(kf.rgbd)[(kf.frame_to_store[c] * kf.height + ic) * (int64_t)kf.width + jc];
  const float gt = px.w / (1.0f / nrm);
  const bool vd = gt != 0.0f;
  o.near[idx] = nearv; o.far[idx] = farv; o.gt[idx] = gt;
  reinterpret_cast<float4*>(o.rgbds)[idx] = px;
  o.rgb_mask[idx] = (px.x != 0.f || px.y != 0.f) ? 1 : 0;
  o.depth_mask[idx] = (gt > nearv && gt < farv && vd) ? 1 : 0;
  o.term_probs[idx] = (gt < farv) ? 1.0f : 0.0f;
  o.term_mask[idx] = (gt > nearv && vd) ? 1 : 0;
}

// ---- single-view variant (NeuralGraphMap._sample_target_sv, rm.py:1461-1583) ------------------------------------
// k_target_sv_intersect: does the segment camera origin -> back-projected depth point n pass through the sphere of field f?
//   geometry.LineSegments.closest_points / intersects_spheres (geometry.py:67-105) with p1 = 0: t = clamp(c.p / |p|^2, 0, 1)
//   (|p|^2 == 0 -> 1), closest = p t, hit = |c - closest|^2 <= r^2.  One thread per (field, point); 50 000 points x up
//   to a few hundred fields of byte output: HBM-bound.
__global__ void k_target_sv_intersect(int F, int64_t N, const float* __restrict__ pos_c, const float* __restrict__ points, float radius,
                                      uint8_t* __restrict__ hit) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int f = blockIdx.y;
  if (n >= N) return;
  const float px = points[3 * n], py = points[3 * n + 1], pz = points[3 * n + 2];
  const float cx = pos_c[3 * f], cy = pos_c[3 * f + 1], cz = pos_c[3 * f + 2];
  float sq = (px * px + py * py) + pz * pz;
  if (sq == 0.0f) sq = 1.0f;
  const float t = fminf(fmaxf(((cx * px + cy * py) + cz * pz) / sq, 0.0f), 1.0f);
  const float ex = cx - px * t, ey = cy - py * t, ez = cz - pz * t;
  hit[(int64_t)f * N + n] = ((ex * ex + ey * ey) + ez * ez <= radius * radius) ? 1 : 0;
}
// k_target_sv_rays: per sampled segment: pixel, direction, near / far from the field sphere (NOT clamped at 0 here,
// rm.py:1543-1544), RGB-D target, ray distance of the depth, masks (rm.py:1536-1561)
__global__ void k_target_sv_rays(int F, int R, const float* __restrict__ pos_c, float radius, const int64_t* __restrict__ pts_ijs,
                                 const int64_t* __restrict__ segments, const float* __restrict__ image, int height, int width,
                                 float fx, float fy, float cx, float cy, ngm_target_out o) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)F * R) return;
  const int f = (int)(idx / R);
  const int64_t sgm = segments[idx];
  const int64_t i = pts_ijs[2 * sgm], j = pts_ijs[2 * sgm + 1];
  o.ijs[2 * idx] = i; o.ijs[2 * idx + 1] = j;
  const float dx = ((float)j - cx) / fx, dy = ((float)i - cy) / fy;
  const float nrm = fmaxf(sqrtf(dx * dx + dy * dy + 1.0f), 1e-12f);
  const float gx = dx / nrm, gy = (-dy) / nrm, gz = -1.0f / nrm;
  const float center = pos_c[3 * f] * gx + pos_c[3 * f + 1] * gy + pos_c[3 * f + 2] * gz;
  const float nearv = center - radius, farv = center + radius;
  const float4 px = reinterpret_cast<const float4*>(image)[i * (int64_t)width + j];
  const float gt = px.w / (1.0f / nrm);
  const bool dm = gt < farv;
  o.near[idx] = nearv; o.far[idx] = farv; o.gt[idx] = gt;
  reinterpret_cast<float4*>(o.rgbds)[idx] = px;
  o.rgb_mask[idx] = dm ? 1 : 0;
  o.depth_mask[idx] = dm ? 1 : 0;
  o.term_probs[idx] = dm ? 1.0f : 0.0f;
  o.term_mask[idx] = 1;
  (void)height;
}
int ngm_launch_target_sv_intersect(int F, int64_t N, const float* pos_c, const float* points, float radius, uint8_t* hit, hipStream_t st) {
  hipLaunchKernelGGL(k_target_sv_intersect, dim3((unsigned)((N + 255) / 256), (unsigned)F), dim3(256), 0, st, F, N, pos_c, points, radius, hit);
  return 0;
}
int ngm_launch_target_sv_rays(int F, int R, const float* pos_c, float radius, const int64_t* pts_ijs, const int64_t* segments,
                              const float* image, int height, int width, float fx, float fy, float cx, float cy, const ngm_target_out& o,
                              hipStream_t st) {
  const int64_t n = (int64_t)F * R;
  hipLaunchKernelGGL(k_target_sv_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, F, R, pos_c, radius, pts_ijs, segments, image,
                     height, width, fx, fy, cx, cy, o);
  return 0;
}

int ngm_launch_target_visibility(const ngm_keyframes& kf, int F, const float* field_pos, int num_offsets, const float* offsets,
                                 float radius, uint8_t* kf_mask, float* bbox, hipStream_t st) {
  const int n = F * kf.num_frames;
  hipLaunchKernelGGL(k_target_visibility, dim3((n + 127) / 128), dim3(128), 0, st, kf, F, field_pos, num_offsets, offsets, radius,
                     kf_mask, bbox);
  return 0;
}
int ngm_launch_target_rays(const ngm_keyframes& kf, int F, int R, const float* field_pos, float radius, const float* bbox,
                           const int64_t* frame_cids, const float* u_xy, const ngm_target_out& o, hipStream_t st) {
  const int64_t n = (int64_t)F * R;
  hipLaunchKernelGGL(k_target_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, kf, F, R, field_pos, radius, bbox,
                     frame_cids, u_xy, o);
  return 0;
}
