// Greedy 3-D NMS over score-sorted boxes: the build's counterpart of _contrib_NMS3D (operator_cxx/contrib/nms_3d.cu in
// the reference: geometry :54-368, pair mask :380-431, serial keep loop :433-464, forward :466-534).
//
// The reference materialises the whole N x N/64 suppression mask (313 MB per frame at N = 50 000, times the batch) and
// walks it with one GPU thread.  Here the rows are processed in blocks (64, 128, ... up to NMS3D_RB rows: the first rows
// suppress most of the list, so the dense all-alive part of the work stays small): a pair kernel fills the mask rows of
// the block -- only for rows that are still alive and only against columns that are still alive when the block starts
// (suppression is monotone, so leaving those bits 0 cannot change the result) -- and a scan kernel (one workgroup per
// frame, suppression bitset in LDS) runs the serial keep loop over the block, jumping from kept row to kept row.  The
// mask is NMS3D_RB x N/64 words per frame (6.4 MB at N = 50 000).  Once max_keep rows are kept a flag turns the remaining
// launches into no-ops.  Result = the reference's: keep_idx (-1 padded), bbox_after_nms (0 padded).
//
// Float arithmetic follows the reference's order with FP contraction off (iou > thresh decides keep/suppress).
#pragma once
#include "rd_common.h"

namespace rd {
#define RD_NOCONTRACT_N _Pragma("clang fp contract(off)")
constexpr int NMS3D_RB = 1024;      // rows per block
constexpr int NMS3D_SEG = 2048;     // columns per pair-kernel workgroup (256 threads x 8 iterations)
constexpr float NMS3D_EPS = 1e-8f;  // nms_3d.cu:29

struct N3Pt { float x, y; };

__device__ __forceinline__ float n3_cross2(N3Pt a, N3Pt b) { RD_NOCONTRACT_N return a.x * b.y - a.y * b.x; }                 // :54-56
__device__ __forceinline__ float n3_cross3(N3Pt p1, N3Pt p2, N3Pt p0) {                                                   // :65-67
  RD_NOCONTRACT_N
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
// segment p0->p1 against q0->q1 (:154-183)
__device__ __forceinline__ int n3_meet(N3Pt p1, N3Pt p0, N3Pt q1, N3Pt q0, N3Pt& ans) {
  RD_NOCONTRACT_N
  const bool rc = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
                  fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
  if (!rc) return 0;
  const float s1 = n3_cross3(q0, p1, p0), s2 = n3_cross3(p1, q1, p0);
  const float s3 = n3_cross3(p0, q1, q0), s4 = n3_cross3(q1, p1, q0);
  if (!(s1 * s2 > 0.f && s3 * s4 > 0.f)) return 0;
  const float s5 = n3_cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > NMS3D_EPS) {
    ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    ans.x = (b0 * c1 - b1 * c0) / D;
    ans.y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}
// point inside the quadrilateral, with the reference's -1e-2 margin on the signed cross products (:95-152)
__device__ __forceinline__ int n3_inside(const N3Pt* c, N3Pt P) {
  RD_NOCONTRACT_N
  const float margin = -1e-2f;
  const N3Pt ab = {c[1].x - c[0].x, c[1].y - c[0].y}, bc = {c[2].x - c[1].x, c[2].y - c[1].y};
  const N3Pt cd = {c[3].x - c[2].x, c[3].y - c[2].y}, da = {c[0].x - c[3].x, c[0].y - c[3].y};
  const float cw = n3_cross2(ab, bc);
  const N3Pt pa = {c[0].x - P.x, c[0].y - P.y};
  if (n3_cross2(pa, ab) * cw < margin) return 0;
  const N3Pt pb = {c[1].x - P.x, c[1].y - P.y};
  if (n3_cross2(pb, bc) * cw < margin) return 0;
  const N3Pt pc = {c[2].x - P.x, c[2].y - P.y};
  if (n3_cross2(pc, cd) * cw < margin) return 0;
  const N3Pt pd = {c[3].x - P.x, c[3].y - P.y};
  if (n3_cross2(pd, da) * cw < margin) return 0;
  return 1;
}
__device__ __forceinline__ float n3_area(const float* b) {   // :195-200
  RD_NOCONTRACT_N
  const float e1 = (b[0] - b[2]) * (b[0] - b[2]) + (b[1] - b[3]) * (b[1] - b[3]);
  const float e2 = (b[4] - b[2]) * (b[4] - b[2]) + (b[5] - b[3]) * (b[5] - b[3]);
  return sqrtf(e1 * e2);
}
// area of the intersection polygon of two quadrilaterals (:220-340)
__device__ float n3_overlap(const float* a, const float* b) {
  RD_NOCONTRACT_N
  N3Pt ac[5], bc[5];
#pragma unroll
  for (int k = 0; k < 4; ++k) { ac[k] = {a[2 * k], a[2 * k + 1]}; bc[k] = {b[2 * k], b[2 * k + 1]}; }
  ac[4] = ac[0];
  bc[4] = bc[0];
  N3Pt cp[16];
  N3Pt ctr = {0.f, 0.f};
  int cnt = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      N3Pt t;
      if (n3_meet(ac[i + 1], ac[i], bc[j + 1], bc[j], t)) { ctr.x = ctr.x + t.x; ctr.y = ctr.y + t.y; cp[cnt++] = t; }
    }
  for (int k = 0; k < 4; k++) {
    if (n3_inside(ac, bc[k])) { ctr.x = ctr.x + bc[k].x; ctr.y = ctr.y + bc[k].y; cp[cnt++] = bc[k]; }
    if (n3_inside(bc, ac[k])) { ctr.x = ctr.x + ac[k].x; ctr.y = ctr.y + ac[k].y; cp[cnt++] = ac[k]; }
  }
  ctr.x /= cnt;
  ctr.y /= cnt;
  for (int j = 0; j < cnt - 1; j++)
    for (int i = 0; i < cnt - j - 1; i++)
      if (atan2f(cp[i].y - ctr.y, cp[i].x - ctr.x) > atan2f(cp[i + 1].y - ctr.y, cp[i + 1].x - ctr.x)) {
        N3Pt t = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = t;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; k++) {
    const N3Pt u = {cp[k].x - cp[0].x, cp[k].y - cp[0].y};
    const N3Pt v = {cp[k + 1].x - cp[0].x, cp[k + 1].y - cp[0].y};
    area += n3_cross2(u, v);
  }
  return (float)((double)fabsf(area) / 2.0);
}
__device__ __forceinline__ float n3_iou_bev(const float* a, const float* b) {   // :342-368 (a volume ratio, despite the name)
  RD_NOCONTRACT_N
  const float ha = a[9] - a[8], hb = b[9] - b[8];
  float oh = fminf(a[9], b[9]) - fmaxf(a[8], b[8]);
  if (oh < 0.f) oh = 0.f;
  const float va = n3_area(a) * ha, vb = n3_area(b) * hb;
  const float vo = n3_overlap(a, b) * oh;
  return vo / fmaxf(va + vb - vo, NMS3D_EPS);
}
__device__ __forceinline__ float n3_iou_normal(const float* a, const float* b) {   // :370-378, boxes read as x1,y1,x2,y2
  RD_NOCONTRACT_N
  const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
  const float inter = w * h;
  const float sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
  return inter / fmaxf(sa + sb - inter, NMS3D_EPS);
}

// per-frame state in the workspace: [0] rows kept so far, [1] done flag
struct Nms3dFrame {
  unsigned long long* remv;   // ncw words: bit j = box j is suppressed
  unsigned long long* mask;   // NMS3D_RB x ncw words
  int* state;
};
struct Nms3dArgs {
  const float* boxes;         // (B, N, 10)
  unsigned char* ws;
  size_t ws_frame;            // bytes per frame in the workspace
  int* keep;                  // (B, max_keep)
  float* out;                 // (B, max_keep, 10)
  int N, ncw, max_keep, normal_iou;
  float thresh;
};
__host__ __device__ inline size_t nms3d_frame_bytes(int N) {
  const size_t ncw = (size_t)(N + 63) / 64;
  return ((ncw * 8 + 255) & ~(size_t)255) + (size_t)NMS3D_RB * ncw * 8 + 256;
}
__device__ __forceinline__ Nms3dFrame nms3d_frame(const Nms3dArgs& a, int b) {
  unsigned char* p = a.ws + (size_t)b * a.ws_frame;
  const size_t ncw = (size_t)a.ncw;
  Nms3dFrame f;
  f.remv = (unsigned long long*)p;
  f.mask = (unsigned long long*)(p + ((ncw * 8 + 255) & ~(size_t)255));
  f.state = (int*)(p + ((ncw * 8 + 255) & ~(size_t)255) + (size_t)NMS3D_RB * ncw * 8);
  return f;
}

__global__ __launch_bounds__(256) void nms3d_init_kernel(Nms3dArgs a) {
  const Nms3dFrame f = nms3d_frame(a, blockIdx.y);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < a.ncw) {                                   // columns past N count as suppressed
    const int lo = i * 64;
    f.remv[i] = a.N - lo >= 64 ? 0ull : ~0ull << (a.N - lo);
  }
  if (i < 2) f.state[i] = 0;
  if (i < a.max_keep) a.keep[(size_t)blockIdx.y * a.max_keep + i] = -1;
  for (int k = i; k < a.max_keep * 10; k += gridDim.x * 256) a.out[(size_t)blockIdx.y * a.max_keep * 10 + k] = 0.f;
}

// grid (ceil(N / NMS3D_SEG), NMS3D_RB, B): mask words of row r0 + blockIdx.y for the columns of segment blockIdx.x
__global__ __launch_bounds__(256) void nms3d_pairs_kernel(Nms3dArgs a, int r0) {
  const Nms3dFrame f = nms3d_frame(a, blockIdx.z);
  const int i = r0 + blockIdx.y;
  const int seg0 = blockIdx.x * NMS3D_SEG;
  if (i >= a.N || seg0 + NMS3D_SEG <= (i & ~63)) return;                   // every word of the segment is left of the row's own word
  if (f.state[1]) return;
  if ((f.remv[i >> 6] >> (i & 63)) & 1ull) return;                         // the row is already suppressed
  const float* bx = a.boxes + (size_t)blockIdx.z * a.N * 10;
  float bi[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) bi[k] = bx[(size_t)i * 10 + k];
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < NMS3D_SEG / 256; ++it) {
    const int j0 = seg0 + it * 256 + (threadIdx.x & ~63);                  // first column of this wave's word
    if (j0 >= a.N) break;
    const int cw = j0 >> 6;
    if (cw < (i >> 6)) continue;                                           // left of the row's own word: never read by the scan
    const unsigned long long dead = f.remv[cw];
    unsigned long long bits = 0ull;
    if (dead != ~0ull) {
      const int j = j0 + lane;
      bool hit = false;
      if (j > i && j < a.N && !((dead >> lane) & 1ull)) {
        float bj[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) bj[k] = bx[(size_t)j * 10 + k];
        const float v = a.normal_iou ? n3_iou_normal(bi, bj) : n3_iou_bev(bi, bj);
        hit = v > a.thresh;
      }
      bits = __ballot(hit);
    }
    if (lane == 0) f.mask[(size_t)blockIdx.y * a.ncw + cw] = bits;
  }
}

// grid (B): the serial keep loop over rows [r0, r0 + rows), suppression bitset in LDS (r0 and rows are multiples of 64)
__global__ __launch_bounds__(256) void nms3d_scan_kernel(Nms3dArgs a, int r0, int rows) {
  HIP_DYNAMIC_SHARED(unsigned long long, remv);
  const Nms3dFrame f = nms3d_frame(a, blockIdx.x);
  if (f.state[1]) return;
  const int tid = threadIdx.x;
  for (int w = tid; w < a.ncw; w += 256) remv[w] = f.remv[w];
  __syncthreads();
  int nk = f.state[0];
  const float* bx = a.boxes + (size_t)blockIdx.x * a.N * 10;
  const int r1 = min(r0 + rows, a.N);
  bool done = false;
  for (int w = r0 >> 6; w < (r1 + 63) >> 6 && !done; ++w) {
    int from = 0;                                      // rows of this word below `from` are settled
    while (true) {
      const unsigned long long alive = ~remv[w] & (from >= 64 ? 0ull : ~0ull << from);
      if (!alive) break;
      const int bit = __ffsll((unsigned long long)alive) - 1;
      const int i = w * 64 + bit;                      // uniform: every thread reads the same LDS word
      if (tid < 10) a.out[((size_t)blockIdx.x * a.max_keep + nk) * 10 + tid] = bx[(size_t)i * 10 + tid];
      if (tid == 10) a.keep[(size_t)blockIdx.x * a.max_keep + nk] = i;
      ++nk;
      if (nk >= a.max_keep) { done = true; break; }
      const unsigned long long* row = f.mask + (size_t)(i - r0) * a.ncw;
      __syncthreads();                                 // everyone has read remv[w] for this step
      for (int c = w + tid; c < a.ncw; c += 256) remv[c] |= row[c];
      __syncthreads();
      from = bit + 1;
    }
  }
  __syncthreads();
  for (int w = tid; w < a.ncw; w += 256) f.remv[w] = remv[w];
  if (tid == 0) {
    f.state[0] = nk;
    if (done || r1 >= a.N) f.state[1] = 1;
  }
}
}  // namespace rd
