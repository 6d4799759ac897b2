// 8-point rotated IoU (operator_cxx/contrib/rotated_iou-inl.h:49-128,130-172,186-192,388-493 in the reference)
// and the 'bev' batch_rotated_iou reduction (operator_py/batch_rotated_iou.py:33-49): one thread per box pair,
// float arithmetic in the reference's order (FP contraction off).
#pragma once
#include "rd_common.h"

namespace rd {
struct RPt { float x, y; };
#define RD_NOCONTRACT_R _Pragma("clang fp contract(off)")

__device__ __forceinline__ bool r_eq(float d1, float d2) {  // :50-53, EPS 1e-8, divides by min(d1,d2)
  RD_NOCONTRACT_R
  float m = d1 < d2 ? d1 : d2;
  return fabsf((d1 - d2) / m) < 1e-8f;
}
__device__ __forceinline__ int r_inside(const float* box, RPt p) {  // :112-128
  RD_NOCONTRACT_R
  int flag = -1;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int j = (i + 1) & 3;
    float pos = (box[2 * j] - box[2 * i]) * (p.y - box[2 * i + 1]) - (box[2 * j + 1] - box[2 * i + 1]) * (p.x - box[2 * i]);
    int ge = pos >= 0.0f;
    if (flag == -1) flag = ge;
    else if (flag != ge) return 0;
  }
  return 1;
}
__device__ __forceinline__ int r_meet(RPt p1, RPt p0, RPt q1, RPt q0, RPt& ans) {  // :130-172
  RD_NOCONTRACT_R
  bool rc = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
            fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
  if (!rc) return 0;
  float A1 = p1.y - p0.y, B1 = p0.x - p1.x, C1 = A1 * p0.x + B1 * p0.y;
  float A2 = q1.y - q0.y, B2 = q0.x - q1.x, C2 = A2 * q0.x + B2 * q0.y;
  float det = A1 * B2 - A2 * B1;
  if (r_eq(det, 0.0f)) return 0;
  float x = (B2 * C1 - B1 * C2) / det;
  float y = (A1 * C2 - A2 * C1) / det;
  float lx = fminf(p0.x, p1.x), hx = fmaxf(p0.x, p1.x), ly = fminf(p0.y, p1.y), hy = fmaxf(p0.y, p1.y);
  bool on1 = (lx < x || r_eq(lx, x)) && (hx > x || r_eq(hx, x)) && (ly < y || r_eq(ly, y)) && (hy > y || r_eq(hy, y));
  lx = fminf(q0.x, q1.x); hx = fmaxf(q0.x, q1.x); ly = fminf(q0.y, q1.y); hy = fmaxf(q0.y, q1.y);
  bool on2 = (lx < x || r_eq(lx, x)) && (hx > x || r_eq(hx, x)) && (ly < y || r_eq(ly, y)) && (hy > y || r_eq(hy, y));
  if (on1 && on2) { ans.x = x; ans.y = y; return 1; }
  return 0;
}
__device__ float r_iou8(const float* a, const float* b) {  // :388-464, :477-493
  RD_NOCONTRACT_R
  float sa = (a[2] - a[0]) * (a[5] - a[1]) - (a[3] - a[1]) * (a[4] - a[0]);
  sa += (a[4] - a[0]) * (a[7] - a[1]) - (a[5] - a[1]) * (a[6] - a[0]);
  float sb = (b[2] - b[0]) * (b[5] - b[1]) - (b[3] - b[1]) * (b[4] - b[0]);
  sb += (b[4] - b[0]) * (b[7] - b[1]) - (b[5] - b[1]) * (b[6] - b[0]);
  sa = (float)((double)fabsf(sa) / 2.0);
  sb = (float)((double)fabsf(sb) / 2.0);
  if (sa < 1e-8f || sb < 1e-8f) return 0.0f;
  RPt ac[5], bc[5];
#pragma unroll
  for (int k = 0; k < 4; ++k) { ac[k] = {a[2 * k], a[2 * k + 1]}; bc[k] = {b[2 * k], b[2 * k + 1]}; }
  ac[4] = ac[0];
  bc[4] = bc[0];
  RPt cp[16];
  RPt ctr = {0.f, 0.f};
  int cnt = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      RPt t;
      if (r_meet(ac[i + 1], ac[i], bc[j + 1], bc[j], t)) { cp[cnt] = t; ctr.x = ctr.x + t.x; ctr.y = ctr.y + t.y; cnt++; }
    }
  for (int k = 0; k < 4; k++) {
    if (r_inside(a, bc[k])) { ctr.x = ctr.x + bc[k].x; ctr.y = ctr.y + bc[k].y; cp[cnt++] = bc[k]; }
    if (r_inside(b, ac[k])) { ctr.x = ctr.x + ac[k].x; ctr.y = ctr.y + ac[k].y; cp[cnt++] = ac[k]; }
  }
  ctr.x /= cnt;
  ctr.y /= cnt;
  for (int j = 0; j < cnt - 1; j++)
    for (int i = 0; i < cnt - j - 1; i++)
      if (atan2f(cp[i].y - ctr.y, cp[i].x - ctr.x) > atan2f(cp[i + 1].y - ctr.y, cp[i + 1].x - ctr.x)) {
        RPt t = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = t;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; k++) {
    RPt u = {cp[k].x - cp[0].x, cp[k].y - cp[0].y};
    RPt v = {cp[k + 1].x - cp[0].x, cp[k + 1].y - cp[0].y};
    area += u.x * v.y - u.y * v.x;
  }
  float s = (float)((double)fabsf(area) / 2.0);
  return s / fmaxf(sa + sb - s, 1e-8f);
}
// ---- 7-dim boxes [x, y, z, w, l, h, angle]: volume IoU (rotated_iou-inl.h:96-110,174-184,284-386,495-507) ----------------------
__device__ __forceinline__ int r_inside7(const float* box, RPt p) {  // check_in_box2d_xyzwlh :96-110
  RD_NOCONTRACT_R
  float ac = cosf(-box[6]), as = sinf(-box[6]);
  float rx = (p.x - box[0]) * ac + (p.y - box[1]) * as + box[0];
  float ry = -(p.x - box[0]) * as + (p.y - box[1]) * ac + box[1];
  return rx >= box[0] - box[3] / 2 && rx <= box[0] + box[3] / 2 && ry >= box[1] - box[4] / 2 && ry <= box[1] + box[4] / 2;
}
__device__ __forceinline__ void r_corners7(const float* box, RPt* c) {  // :294-323, rotate_around_center :174-184
  RD_NOCONTRACT_R
  const float x = box[0], y = box[1], w = box[3], l = box[4];
  c[0] = {x - w / 2, y - l / 2};
  c[1] = {x + w / 2, y - l / 2};
  c[2] = {x + w / 2, y + l / 2};
  c[3] = {x - w / 2, y + l / 2};
  const float ac = cosf(box[6]), as = sinf(box[6]);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float nx = (c[k].x - x) * ac + (c[k].y - y) * as + x;
    const float ny = -(c[k].x - x) * as + (c[k].y - y) * ac + y;
    c[k] = {nx, ny};
  }
  c[4] = c[0];
}
// volume IoU of two 7-dim boxes whose rotated corners ac / bc (5 points, closed) are already known
__device__ float r_iou7c(const float* a, const float* b, const RPt* ac, const RPt* bc) {  // :284-386, :495-507
  RD_NOCONTRACT_R
  const float sa = a[3] * a[4] * a[5], sb = b[3] * b[4] * b[5];
  if (sa < 1e-8f || sb < 1e-8f) return 0.0f;
  RPt cp[16];
  RPt ctr = {0.f, 0.f};
  int cnt = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      RPt t;
      if (r_meet(ac[i + 1], ac[i], bc[j + 1], bc[j], t)) { cp[cnt] = t; ctr.x = ctr.x + t.x; ctr.y = ctr.y + t.y; cnt++; }
    }
  for (int k = 0; k < 4; k++) {
    if (r_inside7(a, bc[k])) { ctr.x = ctr.x + bc[k].x; ctr.y = ctr.y + bc[k].y; cp[cnt++] = bc[k]; }
    if (r_inside7(b, ac[k])) { ctr.x = ctr.x + ac[k].x; ctr.y = ctr.y + ac[k].y; cp[cnt++] = ac[k]; }
  }
  ctr.x /= cnt;
  ctr.y /= cnt;
  for (int j = 0; j < cnt - 1; j++)
    for (int i = 0; i < cnt - j - 1; i++)
      if (atan2f(cp[i].y - ctr.y, cp[i].x - ctr.x) > atan2f(cp[i + 1].y - ctr.y, cp[i + 1].x - ctr.x)) {
        RPt t = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = t;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; k++) {
    RPt u = {cp[k].x - cp[0].x, cp[k].y - cp[0].y};
    RPt v = {cp[k + 1].x - cp[0].x, cp[k + 1].y - cp[0].y};
    area += u.x * v.y - u.y * v.x;
  }
  const float s = (float)((double)fabsf(area) / 2.0);
  const float top = fminf(a[2] + a[5] / 2.0f, b[2] + b[5] / 2.0f), bot = fmaxf(a[2] - a[5] / 2.0f, b[2] - b[5] / 2.0f);
  const float h = fmaxf(0.0f, top - bot);
  return s * h / fmaxf(sa + sb - s * h, 1e-8f);
}
__global__ __launch_bounds__(256) void riou7_kernel(const float* __restrict__ b1, const float* __restrict__ b2,
                                                    float* __restrict__ out, long n1, long n2) {
  long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= n1 * n2) return;
  long r = i / n2, c = i - r * n2;
  RPt ac[5], bc[5];
  r_corners7(b1 + r * 7, ac);
  r_corners7(b2 + c * 7, bc);
  out[i] = r_iou7c(b1 + r * 7, b2 + c * 7, ac, bc);
}
// Custom op 'batch_rotated_iou' with iou_type '3d' (operator_py/batch_rotated_iou.py:17-18,36-39,51-68): proposals (B,N,10) are
// converted to [cx, cy, cz, length, width, height, yaw] (to_box_type_7), the yaw of proposals AND ground truth (B,n_gt,7) is negated,
// volume IoU, cleaned, maximum over the frame's ground-truth boxes.  Same structure as the 'bev' kernel below: the frame's GT boxes,
// their rotated corners and bounds in LDS, one thread per proposal.  A pair whose corner bounds are apart by more than a rounding
// margin has no edge crossing (every rectangle test of :131-136 fails) and no corner inside the other box, so the reference gets
// cnt = 0, area 0 and returns exactly 0: not clipped.
__global__ __launch_bounds__(256) void batch_riou3d_kernel(const float* __restrict__ prop, int pstride, long prop_bs,
                                                           const float* __restrict__ gt, long gt_bs, float* __restrict__ out,
                                                           int* __restrict__ out_arg, long n, int ngt) {
  RD_NOCONTRACT_R
  __shared__ float g[256 * 7];
  __shared__ float gc[256 * 8];   // rotated corners
  __shared__ float gb[256 * 4];   // min x, max x, min y, max y
  const int b = blockIdx.y;
  gt += b * gt_bs;
  for (int i = threadIdx.x; i < ngt * 7; i += 256) g[i] = (i % 7 == 6) ? -1 * gt[i] : gt[i];   // gt_batch[:, -1] = -1 * gt_batch[:, -1]
  __syncthreads();
  for (int j = threadIdx.x; j < ngt; j += 256) {
    RPt c[5];
    r_corners7(g + j * 7, c);
    float lx = c[0].x, hx = c[0].x, ly = c[0].y, hy = c[0].y;
    bool nan = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gc[8 * j + 2 * k] = c[k].x; gc[8 * j + 2 * k + 1] = c[k].y;
      lx = fminf(lx, c[k].x); hx = fmaxf(hx, c[k].x); ly = fminf(ly, c[k].y); hy = fmaxf(hy, c[k].y);
      nan |= !(c[k].x == c[k].x) || !(c[k].y == c[k].y);
    }
    if (nan) { lx = -INFINITY; hx = INFINITY; ly = -INFINITY; hy = INFINITY; }
    gb[4 * j] = lx; gb[4 * j + 1] = hx; gb[4 * j + 2] = ly; gb[4 * j + 3] = hy;
  }
  __syncthreads();
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= n) return;
  const float* p = prop + b * prop_bs + i * pstride;
  float box[7];   // to_box_type_7 (:51-68), then roi_batch[:, -1] = -1 * roi_batch[:, -1]
  box[0] = (((p[0] + p[2]) + p[4]) + p[6]) / 4.0f;
  box[1] = (((p[1] + p[3]) + p[5]) + p[7]) / 4.0f;
  box[2] = (p[8] + p[9]) / 2.0f;
  box[3] = sqrtf((p[0] - p[2]) * (p[0] - p[2]) + (p[1] - p[3]) * (p[1] - p[3]));
  box[4] = sqrtf((p[2] - p[4]) * (p[2] - p[4]) + (p[3] - p[5]) * (p[3] - p[5]));
  box[5] = p[9] - p[8];
  box[6] = -1 * atan2f(p[1] - p[3], p[0] - p[2]);
  RPt ac[5];
  r_corners7(box, ac);
  float lx = ac[0].x, hx = ac[0].x, ly = ac[0].y, hy = ac[0].y;
  bool finite = true;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    lx = fminf(lx, ac[k].x); hx = fmaxf(hx, ac[k].x); ly = fminf(ly, ac[k].y); hy = fmaxf(hy, ac[k].y);
    finite &= ac[k].x == ac[k].x && ac[k].y == ac[k].y;
  }
  // rounding margin: the in-box test (:96-110) rotates the POINT back instead of comparing with the rotated corners
  const float mg = 1e-4f * (1.f + fabsf(lx) + fabsf(hx) + fabsf(ly) + fabsf(hy));
  float best = -1.f;
  int arg = 0;
  for (int j = 0; j < ngt; ++j) {
    float v = 0.f;
    const bool apart = hx + mg < gb[4 * j] || gb[4 * j + 1] + mg < lx || hy + mg < gb[4 * j + 2] || gb[4 * j + 3] + mg < ly;
    if (!(finite && apart)) {
      RPt bc[5];
#pragma unroll
      for (int k = 0; k < 4; ++k) bc[k] = {gc[8 * j + 2 * k], gc[8 * j + 2 * k + 1]};
      bc[4] = bc[0];
      v = r_iou7c(box, g + j * 7, ac, bc);
      if (!(v == v) || isinf(v) || v > 1.0f || v < 0.f) v = 0.f;
    }
    if (v > best) { best = v; arg = j; }
  }
  out[b * n + i] = best;
  if (out_arg) out_arg[b * n + i] = arg;
}
__global__ __launch_bounds__(256) void riou8_kernel(const float* __restrict__ b1, const float* __restrict__ b2,
                                                    float* __restrict__ out, long n1, long n2) {
  long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= n1 * n2) return;
  long r = i / n2, c = i - r * n2;
  out[i] = r_iou8(b1 + r * 8, b2 + c * 8);
}
// Custom op 'batch_rotated_iou' ('bev', operator_py/batch_rotated_iou.py:11-49): per frame, per proposal the maximum over the
// frame's ground-truth boxes of the cleaned IoU (NaN / Inf / > 1 / < 0 -> 0), and (optionally) the first index reaching it.
// The GT boxes of the frame (<= 256 x 8 floats, the config pads to 200) and their axis-aligned bounds live in LDS; proposals
// stream through, one thread each.  Pairs whose bounds are strictly disjoint are not clipped: for them every edge pair
// fails the routine's own bounding-rectangle test (:131-136) and no corner lies inside the other box, so the reference
// computes cnt = 0, area = 0 and returns exactly 0 (or hits the degenerate-area early return, also 0) -- skipping them is
// result-neutral, bit for bit, and it removes > 99 % of the 34 M pairs of a 169 984 x 200 frame.
__global__ __launch_bounds__(256) void batch_riou_kernel(const float* __restrict__ prop, int pstride, long prop_bs,
                                                         const float* __restrict__ gt, long gt_bs, float* __restrict__ out,
                                                         int* __restrict__ out_arg, long n, int ngt) {
  __shared__ float g[256 * 8];
  __shared__ float gb[256 * 4];   // min x, max x, min y, max y
  const int b = blockIdx.y;
  gt += b * gt_bs;
  for (int i = threadIdx.x; i < ngt * 8; i += 256) g[i] = gt[i];
  __syncthreads();
  for (int j = threadIdx.x; j < ngt; j += 256) {
    const float* q = g + j * 8;
    gb[4 * j + 0] = fminf(fminf(q[0], q[2]), fminf(q[4], q[6]));
    gb[4 * j + 1] = fmaxf(fmaxf(q[0], q[2]), fmaxf(q[4], q[6]));
    gb[4 * j + 2] = fminf(fminf(q[1], q[3]), fminf(q[5], q[7]));
    gb[4 * j + 3] = fmaxf(fmaxf(q[1], q[3]), fmaxf(q[5], q[7]));
    // fminf / fmaxf drop NaN operands: a box with ANY NaN coordinate gets bounds that are never "apart" (so the full routine runs
    // for it, as the reference does for every pair)
    bool nan = false;
#pragma unroll
    for (int c = 0; c < 8; ++c) nan |= !(q[c] == q[c]);
    if (nan) { gb[4 * j + 0] = -INFINITY; gb[4 * j + 1] = INFINITY; gb[4 * j + 2] = -INFINITY; gb[4 * j + 3] = INFINITY; }
  }
  __syncthreads();
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= n) return;
  const float* p = prop + b * prop_bs + i * pstride;
  float box[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) box[k] = p[k];
  const float lx = fminf(fminf(box[0], box[2]), fminf(box[4], box[6])), hx = fmaxf(fmaxf(box[0], box[2]), fmaxf(box[4], box[6]));
  const float ly = fminf(fminf(box[1], box[3]), fminf(box[5], box[7])), hy = fmaxf(fmaxf(box[1], box[3]), fmaxf(box[5], box[7]));
  bool finite = true;                            // ANY NaN coordinate (fminf / fmaxf would drop it): no shortcut, run the routine
#pragma unroll
  for (int k = 0; k < 8; ++k) finite &= box[k] == box[k];
  float best = -1.f;
  int arg = 0;
  for (int j = 0; j < ngt; ++j) {
    float v = 0.f;
    const bool apart = hx < gb[4 * j] || gb[4 * j + 1] < lx || hy < gb[4 * j + 2] || gb[4 * j + 3] < ly;
    if (!(finite && apart)) {
      v = r_iou8(box, g + j * 8);
      if (!(v == v) || isinf(v) || v > 1.0f || v < 0.f) v = 0.f;
    }
    if (v > best) { best = v; arg = j; }          // first maximum, like numpy's argmax
  }
  out[b * n + i] = best;
  if (out_arg) out_arg[b * n + i] = arg;
}
}  // namespace rd
