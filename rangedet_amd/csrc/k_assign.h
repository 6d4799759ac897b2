// Point -> ground-truth box assignment and points-per-box counts: the build's counterpart of processing_cxx.assign3D_v2
// and get_point_num (operator_cxx/src_cxx/assigner.h:11-85 and :87-109 in the reference; callers
// rangedet/core/input.py:293-320,432-438 and util_func.py:56-65).  The reference loops over the 64 x 2650 points on one
// host thread; here one thread owns one point and the boxes sit in LDS.  Float comparisons in the reference's order, FP
// contraction off, so the assignment is index-exact.
#pragma once
#include "rd_common.h"

namespace rd {
#define RD_NOCONTRACT_A _Pragma("clang fp contract(off)")
constexpr int ASSIGN_MAX_BOXES = 1024;
constexpr int POINT_NUM_MAX_BOXES = 500;   // MAX_BOX_NUM, assigner.h:92

struct AssignArgs {
  const float* pc;        // (N,3)
  const float* bbox;      // (M,24): corners A B C D (bottom) E F G H (top), xyz each
  const float* center;    // (M,3)
  const float* radius;    // (M)
  const float* mask;      // (N)
  const float* nlz;       // (N)
  int* out;               // (N)
  long N;
  int M;
  float max_x, min_x, max_y, min_y, max_z, min_z, max_dist;
};

// LDS per box: A.x A.y A.z  B.x B.y  C.x C.y  D.x D.y  E.z  cx cy cz  radius
__global__ __launch_bounds__(256) void assign3d_kernel(AssignArgs a) {
  RD_NOCONTRACT_A
  HIP_DYNAMIC_SHARED(float, bx);
  for (int j = threadIdx.x; j < a.M; j += 256) {
    const float* b = a.bbox + (size_t)j * 24;
    float* d = bx + j * 14;
    d[0] = b[0]; d[1] = b[1]; d[2] = b[2];
    d[3] = b[3]; d[4] = b[4];
    d[5] = b[6]; d[6] = b[7];
    d[7] = b[9]; d[8] = b[10];
    d[9] = b[14];
    d[10] = a.center[j * 3]; d[11] = a.center[j * 3 + 1]; d[12] = a.center[j * 3 + 2];
    d[13] = a.radius[j];
  }
  __syncthreads();
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= a.N) return;
  int res = -1;
  const float px = a.pc[i * 3], py = a.pc[i * 3 + 1], pz = a.pc[i * 3 + 2];
  bool live = !(a.mask[i] < 0.5f || a.nlz[i] > 0.f);                                   // :42
  live = live && !(px < a.min_x || px > a.max_x) && !(py < a.min_y || py > a.max_y) && !(pz < a.min_z || pz > a.max_z);
  if (live) {
    // squared distance to every centre, summed the way Eigen's unrolled 3-element reduction does: x^2 + (y^2 + z^2)  (:47)
    float best = 0.f;
    for (int j = 0; j < a.M; ++j) {
      const float* d = bx + j * 14;
      const float dx = d[10] - px, dy = d[11] - py, dz = d[12] - pz;
      const float q = dx * dx + (dy * dy + dz * dz);
      best = j == 0 ? q : fminf(best, q);
    }
    if (!(best > a.max_dist)) {                                                           // :49 (squared distance vs max_dist)
      for (int j = 0; j < a.M; ++j) {
        const float* d = bx + j * 14;
        const float dx = d[10] - px, dy = d[11] - py, dz = d[12] - pz;
        if (dx * dx + (dy * dy + dz * dz) > d[13]) continue;                            // :51
        if (pz <= d[2] || pz >= d[9]) continue;                                          // :52
        const float ax = d[0], ay = d[1], bxx = d[3], by = d[4], cx = d[5], cy = d[6], ddx = d[7], ddy = d[8];
        if (px < ax && px < bxx && px < cx && px < ddx) continue;
        if (py < ay && py < by && py < cy && py < ddy) continue;
        if (px > ax && px > bxx && px > cx && px > ddx) continue;
        if (py > ay && py > by && py > cy && py > ddy) continue;
        const float bpx = px - bxx, bpy = py - by;
        if ((ax - bxx) * bpx + (ay - by) * bpy <= 0.f) continue;                         // BA . BP
        if ((cx - bxx) * bpx + (cy - by) * bpy <= 0.f) continue;                         // BC . BP
        const float dpx = px - ddx, dpy = py - ddy;
        if ((ax - ddx) * dpx + (ay - ddy) * dpy <= 0.f) continue;                        // DA . DP
        if ((cx - ddx) * dpx + (cy - ddy) * dpy <= 0.f) continue;                        // DC . DP
        res = j;
        break;
      }
    }
  }
  a.out[i] = res;
}

// counts[k] = number of points with index k (float index, truncated like the reference's implicit conversion)
__global__ __launch_bounds__(256) void point_num_count_kernel(const float* __restrict__ inds, long N, int* __restrict__ counts) {
  __shared__ int c[POINT_NUM_MAX_BOXES];
  for (int k = threadIdx.x; k < POINT_NUM_MAX_BOXES; k += 256) c[k] = 0;
  __syncthreads();
  for (long i = blockIdx.x * 256L + threadIdx.x; i < N; i += (long)gridDim.x * 256) {
    const float v = inds[i];
    if (v < 0.f) continue;
    const int k = (int)v;
    if (k < POINT_NUM_MAX_BOXES) atomicAdd(&c[k], 1);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < POINT_NUM_MAX_BOXES; k += 256)
    if (c[k]) atomicAdd(&counts[k], c[k]);
}
__global__ __launch_bounds__(256) void point_num_gather_kernel(const float* __restrict__ inds, long N, const int* __restrict__ counts,
                                                               float* __restrict__ out) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= N) return;
  const float v = inds[i];
  float r = -1.f;
  if (!(v < 0.f)) {
    const int k = (int)v;
    if (k < POINT_NUM_MAX_BOXES) r = (float)counts[k];
  }
  out[i] = r;
}
}  // namespace rd
