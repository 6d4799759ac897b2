// Weighted NMS (operator_cxx/src_cxx/nms.h:452-577 in the reference), decomposed for the GPU so that every
// float operation happens in the reference's order (this file is compiled with FP contraction OFF):
//   1. prep   : per box (in processing order) clockwise-normalised corners, the 4 edge polar angles (the only
//               atan2f of the algorithm -- hoisted out of the O(K^2) part), area, bottom, height.
//   2. pairs  : for sorted positions q1 < q2 the half-plane-intersection IoU (nms.h:96-249) -> two K x K/64
//               bit matrices  thr[q1] (ovr >= thresh)  and  vote[q1] (ovr > thresh_vote).  A pair is evaluated only
//               when the two boxes share a cell of the reference's BBoxHash (nms.h:252-307, used at :470,501,506):
//               the reference never computes the overlap of boxes without a common cell, and that is NOT
//               result-neutral (its clipper returns a non-zero "IoU" for some disjoint boxes), so the predicate is
//               reproduced exactly: per box the cell rectangle [floor(min/s), ceil(max/s)) with the reference's
//               numeric_limits<float>::min() start value of the maxima and its i*100+j key aliasing.
//   3. scan   : one wavefront walks the sorted positions; for an unsuppressed q it snapshots
//               nb[q] = vote[q] & ~supp (written back in place) and ORs thr[q] into supp (64 words per step).
//   4. merge  : one wavefront per kept row: median yaw by rank selection, then the 11 score-weighted sums in
//               neighbourhood order (lane f accumulates field f sequentially -> same rounding as the reference).
// Build note (round 5): this file must be compiled WITHOUT the SLP vectoriser (rangedet_amd/build.py passes
// -fno-slp-vectorize to the device side and lints the code object).  With it, the cross products below become packed-fp32 instructions
// with a swapped second source (v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]); on the MI355X that form returns a wrong low half in
// lanes 48-63 while another wave of the same SIMD issues MFMA instructions, i.e. while this chain overlaps the next batch's convolutions:
// keep counts of a frame then differed by one or two from run to run (prep's area and the pair kernel's overlaps were both hit).
#pragma once
#include "rd_common.h"

namespace rd {

struct WPt { float x, y; };
struct WEdge { WPt a, b; float ang; };
#define RD_NOCONTRACT _Pragma("clang fp contract(off)")
constexpr int PREP_F = 26;  // floats per prepped box: 8 corners, 4 angles, area, bottom, height, [15] rejection-test domain flag,
                            // [16..19] 4 hash-cell bounds (int bits), [20..23] bounding rectangle x0 y0 x1 y1, [24] edge direction mod 90 deg, pad

// ---- rejection test: pairs whose clip is provably irrelevant are not clipped (round 4) -------------------------------------------
// 99 % of the pairs the reference evaluates on a frame's candidates are boxes that do not touch.  The reference's half-plane clipper
// has no empty-intersection test (nms.h:96-149): on disjoint boxes it returns 0, NaN, a negative number or a positive value below
// 1e-8 -- none of which passes `ovr >= thresh` / `ovr > thresh_vote` (nms.h:509-516) -- EXCEPT when two edge directions tie within its
// EPS = 1e-5 (nms.h:58-64,104-106: one of the two half-planes is dropped and the rest can enclose an area: "IoU" up to 1e4), when
// the boxes are nearly parallel (lines meeting thousands of metres away: cancellation in nms.h:54-56,74-90; seen up to 3e-3 rad
// for 40 m boxes, 2e-4 rad up to 20 m) or when the geometry is ill-conditioned (edges of centimetres, coordinates of kilometres).
// Characterised on the reference itself, compiled as-is (oracle/ref_overlap_study.cpp, 8.1e9 disjoint pairs of nine families,
// profiles/r04_nms_spurious_study.txt).  So a pair is skipped -- its overlap taken as 0, exactly what every comparison of the
// reference's value would give -- only when ALL of this holds, with an order of magnitude of margin on each bound:
//   * both boxes are rectangles (1e-3 relative) with edges of 0.2 .. 25 m and |coordinates| <= 200 m       (w_box_domain)
//   * their bounding rectangles are more than 0.01 m apart (then the polygons are disjoint, whatever the rounding)
//   * their edge directions mod 90 degrees differ by at least 0.01 rad
//   * thresh >= 1e-3 and thresh_vote >= 1e-3, BEV mode (the launcher's conditions; the largest reference value seen among skippable
//     pairs is 4.2e-7; the 3-D value divides by a volume sum that was not part of the study, so is3d clips every pair)
// Zero violations on the 8.1e9 pairs; every other pair is clipped as before.  oracle/ and the golden vectors know nothing of this.
__device__ __forceinline__ void w_box_domain(const float* c, float* o) {   // c: the 8 corner floats of a dets row, as given
  RD_NOCONTRACT
  float ex[4], ey[4], l2[4];
  float x0 = c[0], x1 = c[0], y0 = c[1], y1 = c[1];
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int k1 = (k + 1) & 3;
    ex[k] = c[2 * k1] - c[2 * k]; ey[k] = c[2 * k1 + 1] - c[2 * k + 1];
    l2[k] = ex[k] * ex[k] + ey[k] * ey[k];
    ok = ok && l2[k] >= 0.04f && l2[k] <= 625.f && fabsf(c[2 * k]) <= 200.f && fabsf(c[2 * k + 1]) <= 200.f;
    x0 = fminf(x0, c[2 * k]); x1 = fmaxf(x1, c[2 * k]); y0 = fminf(y0, c[2 * k + 1]); y1 = fmaxf(y1, c[2 * k + 1]);
  }
  const float t0 = 1e-3f * sqrtf(l2[0]), t1 = 1e-3f * sqrtf(l2[1]);
  ok = ok && fabsf(ex[0] + ex[2]) <= t0 && fabsf(ey[0] + ey[2]) <= t0 && fabsf(ex[1] + ex[3]) <= t1 && fabsf(ey[1] + ey[3]) <= t1;
  ok = ok && fabsf(ex[0] * ex[1] + ey[0] * ey[1]) <= 1e-3f * sqrtf(l2[0] * l2[1]);
  float a = atan2f(ey[0], ex[0]);                        // direction of one edge, folded to [0, pi/2)
  a = a < 0.f ? a + 3.14159265f : a;
  a = a >= 1.57079633f ? a - 1.57079633f : a;
  o[15] = (ok && a == a) ? 1.f : 0.f;
  o[20] = x0; o[21] = y0; o[22] = x1; o[23] = y1; o[24] = a; o[25] = 0.f;
}
__device__ __forceinline__ bool w_pair_skippable(const float* a, const float* b) {   // a, b: prepped boxes (PREP_F floats)
  RD_NOCONTRACT
  const bool apart = a[22] + 0.01f < b[20] || b[22] + 0.01f < a[20] || a[23] + 0.01f < b[21] || b[23] + 0.01f < a[21];
  float d = fabsf(a[24] - b[24]);
  d = fminf(d, 1.57079633f - d);
  return a[15] != 0.f && b[15] != 0.f && apart && d >= 0.01f;
}

// BBoxHash::getHash (nms.h:268-291): cells (i, j) with i in [c0, c2), j in [c1, c3), key i*100 + j (int).  Two boxes are
// compared iff they share a key (createBBoxMap / getFilterResult, nms.h:256-267,292-303).  i1 - i2 takes every value of
// [a0-b2+1, a2-1-b0] and j2 - j1 every value of [b1-a3+1, b3-1-a1] independently, and the keys collide iff
// 100*(i1-i2) == j2-j1 -- so the test is an interval intersection (the d = 0 term is the plain rectangle overlap).
__device__ __forceinline__ int w_floordiv100(int v) { return v >= 0 ? v / 100 : -((99 - v) / 100); }
__device__ __forceinline__ bool w_share_cell(const int* a, const int* b) {
  if (a[0] >= a[2] || a[1] >= a[3] || b[0] >= b[2] || b[1] >= b[3]) return false;   // a box without cells is in no list
  const int dlo = a[0] - b[2] + 1, dhi = a[2] - 1 - b[0];
  const int jlo = b[1] - a[3] + 1, jhi = b[3] - 1 - a[1];
  const int dmin = -w_floordiv100(-jlo), dmax = w_floordiv100(jhi);       // ceil(jlo/100), floor(jhi/100)
  return max(dlo, dmin) <= min(dhi, dmax);
}


__device__ __forceinline__ int w_sgn(float k) {  // nms.h:48-52, EPS = 1e-5
  RD_NOCONTRACT
  if (fabsf(k) < 1e-5f) return 0;
  return k > 0 ? 1 : -1;
}
__device__ __forceinline__ float w_cross3(WPt o, WPt u, WPt v) {  // nms.h:54-56
  RD_NOCONTRACT
  return (u.x - o.x) * (v.y - o.y) - (u.y - o.y) * (v.x - o.x);
}
__device__ __forceinline__ bool w_less(const WEdge& e1, const WEdge& e2) {  // nms.h:58-64
  RD_NOCONTRACT
  int d = w_sgn(e1.ang - e2.ang);
  if (!d) return w_sgn(w_cross3(e1.a, e2.a, e2.b)) > 0;
  return d < 0;
}
__device__ __forceinline__ WPt w_meet(const WEdge& e1, const WEdge& e2) {  // nms.h:74-83
  RD_NOCONTRACT
  float A1 = e1.b.y - e1.a.y;
  float B1 = e1.a.x - e1.b.x;
  float C1 = (e1.b.x - e1.a.x) * e1.a.y - (e1.b.y - e1.a.y) * e1.a.x;
  float A2 = e2.b.y - e2.a.y;
  float B2 = e2.a.x - e2.b.x;
  float C2 = (e2.b.x - e2.a.x) * e2.a.y - (e2.b.y - e2.a.y) * e2.a.x;
  WPt p;
  p.x = (C2 * B1 - C1 * B2) / (A1 * B2 - A2 * B1);
  p.y = (C1 * A2 - C2 * A1) / (A1 * B2 - A2 * B1);
  return p;
}
__device__ __forceinline__ bool w_outside(const WEdge& e0, const WEdge& e1, const WEdge& e2) {  // nms.h:85-90
  WPt p = w_meet(e1, e2);
  return w_sgn(w_cross3(p, e0.a, e0.b)) > 0;
}

// The 8 directed edges and the deque are indexed dynamically (sort, sweep), which would put them in scratch
// (global memory).  They live in LDS instead, one column per lane: element i of lane t at [i*64 + t] (conflict-free).
struct EdgeLds {
  float *ax, *ay, *bx, *by, *an;  // each [8][64]
  int t;
  __device__ __forceinline__ WEdge get(int i) const {
    WEdge e;
    e.a = {ax[i * 64 + t], ay[i * 64 + t]};
    e.b = {bx[i * 64 + t], by[i * 64 + t]};
    e.ang = an[i * 64 + t];
    return e;
  }
  __device__ __forceinline__ void put(int i, const WEdge& e) const {
    ax[i * 64 + t] = e.a.x; ay[i * 64 + t] = e.a.y;
    bx[i * 64 + t] = e.b.x; by[i * 64 + t] = e.b.y;
    an[i * 64 + t] = e.ang;
  }
};
constexpr int EDGE_LDS_BYTES = 5 * 8 * 64 * 4;

// The reference's sort + de-duplication of the 8 edges, literally (nms.h:98-109): returns the number of edges kept, which then
// sit in positions 0 .. t-1 of the per-lane edge array.
__device__ int w_sort_dedup_literal(const float* pre1, const float* pre2, const EdgeLds& L) {
  RD_NOCONTRACT
  // nms.h:210-225: p[0..3] = box2, p[4..7] = box1 ; l[z] from box2, l[z+4] from box1
#pragma unroll
  for (int z = 0; z < 4; ++z) {
    int z1 = (z + 1) & 3;
    WEdge e;
    e.a = {pre2[2 * z], pre2[2 * z + 1]};
    e.b = {pre2[2 * z1], pre2[2 * z1 + 1]};
    e.ang = pre2[8 + z];
    L.put(z, e);
    e.a = {pre1[2 * z], pre1[2 * z + 1]};
    e.b = {pre1[2 * z1], pre1[2 * z1 + 1]};
    e.ang = pre1[8 + z];
    L.put(z + 4, e);
  }
  // std::sort on 8 elements == libstdc++ __insertion_sort (n <= 16), restated literally because the
  // comparator is not a strict weak order (nms.h:58-64,98)
  for (int i = 1; i < 8; ++i) {
    WEdge val = L.get(i);
    if (w_less(val, L.get(0))) {
      for (int j = i; j > 0; --j) L.put(j, L.get(j - 1));
      L.put(0, val);
    } else {
      int j = i;
      while (w_less(val, L.get(j - 1))) {
        L.put(j, L.get(j - 1));
        --j;
      }
      L.put(j, val);
    }
  }
  int i, j;
  for (i = 0, j = 0; i < 8; i++)
    if (w_sgn(L.an[i * 64 + L.t] - L.an[j * 64 + L.t]) > 0) L.put(++j, L.get(i));
  return j + 1;
}

// box1 = the earlier (kept candidate) box, box2 = the later one; pre1/pre2 their prepped records.
__device__ float w_overlap(const float* pre1, const float* pre2, bool is3d, const EdgeLds& L) {
  RD_NOCONTRACT
  float area1 = pre1[12], area2 = pre2[12];
  // Fast path of the sort + de-duplication below.  When all 28 pairs of edge angles differ by at least EPS, w_less is the plain
  // `<` on the angles -- a strict total order, so ANY sort returns the reference's permutation and the de-duplication loop
  // keeps all 8 edges (every adjacent difference is > EPS).  The sorted position of an edge is then its rank, computed with
  // straight-line code on registers (no divergent loops, no LDS traffic) and each edge is stored once, at its rank.
  // Boxes with (nearly) parallel edges take the literal path.
  int t;
  {
    float ang[8];
#pragma unroll
    for (int z = 0; z < 4; ++z) { ang[z] = pre2[8 + z]; ang[z + 4] = pre1[8 + z]; }
    int rank[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool near = false;
#pragma unroll
    for (int i = 1; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < i; ++j) {
        const float d = ang[i] - ang[j];
        near |= !(fabsf(d) >= 1e-5f);                       // (NaN angles take the literal path too)
        if (d < 0) ++rank[j]; else ++rank[i];
      }
    if (!near) {
#pragma unroll
      for (int z = 0; z < 4; ++z) {
        int z1 = (z + 1) & 3;
        WEdge e;
        e.a = {pre2[2 * z], pre2[2 * z + 1]};
        e.b = {pre2[2 * z1], pre2[2 * z1 + 1]};
        e.ang = pre2[8 + z];
        L.put(rank[z], e);
        e.a = {pre1[2 * z], pre1[2 * z + 1]};
        e.b = {pre1[2 * z1], pre1[2 * z1 + 1]};
        e.ang = pre1[8 + z];
        L.put(rank[z + 4], e);
      }
      t = 8;
    } else {
      t = w_sort_dedup_literal(pre1, pre2, L);
    }
  }
  // the deque of edge positions (values 0..7, at most 10 entries) is a 64-bit register of 4-bit fields, not an LDS column: its
  // reads sit at the head of every dependent chain of the sweep
  unsigned long long dq = 0x10ull;                         // q[0] = 0, q[1] = 1
  auto qg = [&](int k) -> int { return (int)((dq >> (4 * k)) & 15ull); };
  auto qs = [&](int k, int v) { dq = (dq & ~(15ull << (4 * k))) | ((unsigned long long)v << (4 * k)); };
  int top = 1, bot = 0;
  int i;
  for (i = 2; i < t; i++) {
    const WEdge li = L.get(i);
    while (top > bot && w_outside(li, L.get(qg(top)), L.get(qg(top - 1)))) top--;
    while (top > bot && w_outside(li, L.get(qg(bot)), L.get(qg(bot + 1)))) bot++;
    qs(++top, i);
  }
  while (top > bot && w_outside(L.get(qg(bot)), L.get(qg(top)), L.get(qg(top - 1)))) top--;
  while (top > bot && w_outside(L.get(qg(top)), L.get(qg(bot)), L.get(qg(bot + 1)))) bot++;
  {
    const int qb = qg(bot);
    qs(++top, qb);
  }
  // polygon vertices + fan area (nms.h:147-166), vertices generated on the fly
  const int nv = top - bot;
  float inter = 0.f;
  if (nv >= 3) {
    WPt p0 = w_meet(L.get(qg(bot + 1)), L.get(qg(bot)));
    WPt pa = w_meet(L.get(qg(bot + 2)), L.get(qg(bot + 1)));
    float area = 0.f;
    for (int k = 2; k < nv; ++k) {
      WPt pb = w_meet(L.get(qg(bot + k + 1)), L.get(qg(bot + k)));
      area += w_cross3(p0, pa, pb);
      pa = pb;
    }
    if (area < 0) area = -area;
    inter = area / 2;
  }
  if (is3d) {  // nms.h:168-184,234-239
    float bot1 = pre1[13], h1 = pre1[14], top1 = bot1 + h1;
    float bot2 = pre2[13], h2 = pre2[14], top2 = bot2 + h2;
    float min_top = (top1 > top2) ? top2 : top1;
    float max_bot = (bot1 > bot2) ? bot1 : bot2;
    float d = min_top - max_bot;
    float oh = d > 0 ? d : 0;
    inter *= oh;
    area1 *= h1;
    area2 *= h2;
  }
  return inter / (area1 + area2 - inter);
}

// Batched launches: blockIdx.z selects the frame.  WnmsBatch holds the distance between consecutive frames of every
// buffer, in elements of that buffer (0 = shared by all frames, e.g. an identity order).
struct WnmsBatch {
  long dets, order, prep, words, ints, keep, out;   // words: thr / vote / snap;  ints: keep_q;  counts are 1 apart
};
__global__ __launch_bounds__(256) void wnms_prep_kernel(const float* __restrict__ dets, const int* __restrict__ order,
                                                        int cap, const int* __restrict__ d_count, float* __restrict__ prep,
                                                        WnmsBatch bs, float hscale, int* __restrict__ novf) {
  RD_NOCONTRACT
  dets += blockIdx.z * bs.dets; order += blockIdx.z * bs.order; prep += blockIdx.z * bs.prep;
  int q = blockIdx.x * 256 + threadIdx.x;
  if (q == 0) novf[blockIdx.z * bs.ints] = 0;   // overflow list of the merge step starts empty
  int K = d_count ? min(d_count[blockIdx.z], cap) : cap;
  if (q >= K) return;
  const float* b = dets + (size_t)order[q] * 12;
  WPt p[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) p[k] = {b[2 * k], b[2 * k + 1]};
  bool cw = ((p[1].x - p[0].x) * (p[2].y - p[0].y) - (p[2].x - p[0].x) * (p[1].y - p[0].y)) > 0;  // nms.h:92-94
  if (cw) {  // std::reverse of the 4 corners (nms.h:191-193)
    WPt t0 = p[0], t1 = p[1];
    p[0] = p[3];
    p[1] = p[2];
    p[2] = t1;
    p[3] = t0;
  }
  float* o = prep + (size_t)q * PREP_F;
  {  // nms.h:268-286: minima start at max(), maxima at numeric_limits<float>::min() (the smallest POSITIVE normal, not
     // lowest()); y is divided by the x scale for the lower bound and x by the y scale for the upper one (equal here)
    float mn0 = 3.402823466e+38f, mn1 = 3.402823466e+38f, mx0 = 1.175494351e-38f, mx1 = 1.175494351e-38f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mn0 = fminf(mn0, b[2 * k]); mn1 = fminf(mn1, b[2 * k + 1]);
      mx0 = fmaxf(mx0, b[2 * k]); mx1 = fmaxf(mx1, b[2 * k + 1]);
    }
    int c[4] = {0, 0, 1, 1};                               // hscale <= 0: no prefilter, every pair shares cell (0, 0)
    if (hscale > 0.f) {
      c[0] = (short)(int)floorf(mn0 / hscale); c[1] = (short)(int)floorf(mn1 / hscale);
      c[2] = (short)(int)ceilf(mx0 / hscale);  c[3] = (short)(int)ceilf(mx1 / hscale);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) o[16 + k] = __int_as_float(c[k]);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[2 * k] = p[k].x;
    o[2 * k + 1] = p[k].y;
    int k1 = (k + 1) & 3;
    o[8 + k] = fdlibm_atan2f(p[k1].y - p[k].y, p[k1].x - p[k].x);  // nms.h:71 (the C library's atan2f, rd_common.h)
  }
  float area = 0.f;
  area += w_cross3(p[0], p[1], p[2]);
  area += w_cross3(p[0], p[2], p[3]);
  if (area < 0) area = -area;
  o[12] = area / 2;
  o[13] = b[9];
  o[14] = b[10];
  w_box_domain(b, o);
}

// Pair tiles: 64 rows x WN_CT columns per wave -- row block rb (64 consecutive processing positions, or 64 consecutive entries
// of the compacted alive list), column word cb, part `sub` of that word; one wave evaluates the pairs of its tile that the
// reference evaluates and writes WN_CT bits of each row's 64-bit mask word.  The grid is a fixed number of single-wave workgroups
// per frame that stride over the tile list (sized from the DEVICE-side counts, so the launch does not depend on how many boxes
// passed the score filter and a large capacity costs nothing when K is small).
// Round 3: lane-private candidate lists.  Only about a quarter of the pairs of a tile share a BBoxHash cell (the four quadrants
// around the ego vehicle are different cells), and which ones differs from row to row: walking the columns in lock step left
// three of four lanes idle during every polygon clip.  Each lane now first collects the columns IT has to clip (a 32-bit mask:
// later position, not suppressed in round 1, common cell), then all lanes clip their own next candidate together; a lane idles
// only once its list is shorter than the longest of its wave.  Tile width (RD_WNMS_CT): 8 columns = 2 +- 1.2 candidates per lane,
// longest of a wave ~ 5, against 8 lock-step column steps before: batched NMS of 8 frames 793 -> 726 us, +0.8 % frames/s; wider
// tiles balance better (32 columns: 8 +- 2.4, longest ~ 14) but leave the GPU with a quarter of the waves and longer serial
// chains per wave: 16 columns 820 us, 32 columns 1 103 us -- the kernel is bound by the latency of a wave's chain of clips, not
// by lane utilisation.
template <int WN_CT, bool BAL>        // columns per tile: 8 (default), 16 or 32 (RD_WNMS_CT, A/B); BAL: the wave's candidates dealt out evenly
__global__ __launch_bounds__(64) void wnms_pairs_kernel(const float* __restrict__ prep, int cap,
                                                        const int* __restrict__ d_count, float thresh, float thresh_vote,
                                                        int is3d, unsigned long long* __restrict__ thr,
                                                        unsigned long long* __restrict__ vote, int nwcap, WnmsBatch bs,
                                                        const int* __restrict__ rows, const int* __restrict__ nrows,
                                                        const unsigned long long* __restrict__ supp_state, int rb_begin,
                                                        int rb_end, int allow_skip) {
  // rows == nullptr: row blocks [rb_begin, rb_end) of the processing order (first round).  Otherwise (second round) the row
  // blocks are 64 consecutive entries of the compacted, ascending list of rows that survived the first round.
  prep += blockIdx.z * bs.prep; thr += blockIdx.z * bs.words; vote += blockIdx.z * bs.words;
  const int K = d_count ? min(d_count[blockIdx.z], cap) : cap;
  const int ncb = (K + 63) >> 6;
  int nr = K, rb0 = rb_begin, nrb = min(rb_end, ncb) - rb_begin;
  if (rows) {
    rows += blockIdx.z * bs.ints;
    nr = nrows[blockIdx.z * bs.ints];                      // (per-frame workspaces are bs.ints 4-byte words apart)
    rb0 = 0;
    nrb = (nr + 63) >> 6;
    supp_state += blockIdx.z * (bs.ints / 2);
  }
  if (nrb <= 0) return;
  constexpr int WN_SUB = 64 / WN_CT;  // tiles per 64-column mask word
  __shared__ float colp[WN_CT * PREP_F];
  __shared__ float edges[EDGE_LDS_BYTES / 4];
  __shared__ float rowp[BAL ? PREP_F * 64 : 1];              // BAL: the tile's 64 prepped rows, [field][row]
  __shared__ unsigned short plist[BAL ? 64 * WN_CT : 1];     // BAL: candidate pairs (row << 5 | column)
  __shared__ unsigned mbits[BAL ? 128 : 1];                  // BAL: thr / vote bits per row
  const int t = threadIdx.x;
  EdgeLds EL;
  EL.ax = edges; EL.ay = edges + 512; EL.bx = edges + 1024; EL.by = edges + 1536; EL.an = edges + 2048;
  EL.t = t;
  const long ntile = (long)nrb * ncb * WN_SUB;
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int rb = rb0 + (int)(tile / (ncb * WN_SUB)), cb = (int)((tile / WN_SUB) % ncb), sub = (int)(tile % WN_SUB);
    const int qmin = rows ? rows[rb * 64] : rb * 64;
    // skip column tiles that lie entirely before the first row (the tile holding that row itself is still written: the
    // scan and the merge read every row's words from its own, diagonal, word on)
    if (cb * 64 + 63 < qmin) continue;
    // second round: columns of this tile that the first round already suppressed.  Neither their thr bit (ORed into a
    // suppression word that already has it) nor their vote bit (masked by the suppression snapshot in the merge) can
    // influence anything, so those pairs are not evaluated.
    const unsigned dead = rows ? (unsigned)(supp_state[cb] >> (sub * WN_CT)) : 0u;   // (bits >= WN_CT are masked below)
    const int c0 = cb * 64 + sub * WN_CT;
    __syncthreads();                                       // previous tile's colp readers are done
    for (int i = t; i < WN_CT * PREP_F; i += 64) {
      int q2 = c0 + i / PREP_F;
      colp[i] = q2 < K ? prep[(size_t)q2 * PREP_F + (i % PREP_F)] : 0.f;
    }
    __syncthreads();
    const bool active = rb * 64 + t < nr;
    const int q1 = active ? (rows ? rows[rb * 64 + t] : rb * 64 + t) : 0;
    unsigned mt = 0u, mv = 0u, cand = 0u;
    float mine[PREP_F];
    if (active && c0 + WN_CT - 1 > q1) {
#pragma unroll
      for (int k = 0; k < PREP_F; ++k) mine[k] = prep[(size_t)q1 * PREP_F + k];
      int mc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) mc[k] = __float_as_int(mine[16 + k]);
      // this lane's candidates: later in the processing order, inside the frame, alive, and in a common BBoxHash cell
      // (w_share_cell: the reference never evaluates any other pair)
      cand = ~dead;
      if (WN_CT < 32) cand &= (1u << (WN_CT & 31)) - 1u;
      const int lo = q1 + 1 - c0;                          // first column with q2 > q1
      if (lo > 0) cand &= lo >= 32 ? 0u : ~0u << lo;
      const int hi_ = K - c0;                              // columns with q2 < K
      if (hi_ < 32) cand &= hi_ <= 0 ? 0u : (1u << hi_) - 1u;
      for (unsigned rem = cand; rem; rem &= rem - 1u) {
        const int c = __ffs(rem) - 1;
        int oc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) oc[k] = __float_as_int(colp[c * PREP_F + 16 + k]);
        // (round 4) ... and not a pair whose clip cannot pass either threshold (w_pair_skippable above): far apart, well conditioned
        if (!w_share_cell(mc, oc) || (allow_skip && w_pair_skippable(mine, &colp[c * PREP_F]))) cand &= ~(1u << c);
      }
    }
    if constexpr (BAL) {
      // the wave's candidates, compacted and dealt out one per lane and turn: lane t clips pairs t, t + 64, ... of the list,
      // so every lane takes ceil(n / 64) turns whatever its own row's share was.  Rows come from an LDS copy (k-major: the
      // write is conflict-free), result bits are ORed into per-row LDS words.
      if (cand) {
#pragma unroll
        for (int k = 0; k < PREP_F; ++k) rowp[k * 64 + t] = mine[k];
      }
      const int cnt = __popc(cand);
      int incl = cnt;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (t >= d) incl += v;
      }
      const int total = __shfl(incl, 63);
      int pos = incl - cnt;
      for (unsigned rem = cand; rem; rem &= rem - 1u) plist[pos++] = (unsigned short)((t << 5) | (__ffs(rem) - 1));
      mbits[t] = 0u;
      mbits[64 + t] = 0u;
      __syncthreads();
      for (int i = t; i < total; i += 64) {
        const int pr = plist[i], r = pr >> 5, c = pr & 31;
        float a[PREP_F];
#pragma unroll
        for (int k = 0; k < PREP_F; ++k) a[k] = rowp[k * 64 + r];
        const float ovr = w_overlap(a, &colp[c * PREP_F], is3d != 0, EL);
        if (ovr >= thresh) atomicOr(&mbits[r], 1u << c);
        if (ovr > thresh_vote) atomicOr(&mbits[64 + r], 1u << c);
      }
      __syncthreads();
      mt = mbits[t];
      mv = mbits[64 + t];
      if (!active) continue;
    } else {
      if (!active) continue;
      while (cand) {
        const int c = __ffs(cand) - 1;
        cand &= cand - 1u;
        float ovr = w_overlap(mine, &colp[c * PREP_F], is3d != 0, EL);
        if (ovr >= thresh) mt |= 1u << c;
        if (ovr > thresh_vote) mv |= 1u << c;
      }
    }
    typedef typename std::conditional<WN_CT == 32, unsigned, typename std::conditional<WN_CT == 16, unsigned short, unsigned char>::type>::type part_t;
    ((part_t*)thr)[((size_t)q1 * nwcap + cb) * WN_SUB + sub] = (part_t)mt;   // (little endian: part `sub` holds bits WN_CT*sub ..)
    ((part_t*)vote)[((size_t)q1 * nwcap + cb) * WN_SUB + sub] = (part_t)mv;
  }
}

// Greedy scan, one wavefront.  Per 64 sorted rows the needed part of the thr bit-matrix is pulled into LDS with all loads
// in flight at once (one memory latency per 64 rows instead of a dependent load chain per kept row); the rows are then
// resolved sequentially out of LDS.  For every kept row the suppression state BEFORE it is snapshotted to `snap` (the merge
// kernel forms the neighbourhood vote[q] & ~snap), so the vote matrix is never read here.
// The LDS tile holds `tile_w` mask words per row.  K <= 64*tile_w: one column chunk, as described.  Larger K (up to
// RD_WNMS_MAX_K): column chunks of tile_w words -- which rows of a chunk are kept only depends on the chunk's DIAGONAL word
// (column 0 of the first column chunk), so the first column chunk runs the sequential keep loop and records the kept rows,
// and every further column chunk stages just those rows and replays the same snapshot / OR sequence on its own words.
__global__ __launch_bounds__(64) void wnms_scan_kernel(const unsigned long long* __restrict__ thr,
                                                       unsigned long long* __restrict__ snap, int cap,
                                                       const int* __restrict__ d_count, int nwcap,
                                                       const int* __restrict__ order, int* __restrict__ keep_q,
                                                       int* __restrict__ keep, int* __restrict__ d_nkeep, WnmsBatch bs,
                                                       int c_begin, int c_end, unsigned long long* __restrict__ supp_state,
                                                       int tile_w) {
  // 64-row chunks [c_begin, c_end) only.  c_begin > 0 resumes from the suppression state / keep count a previous launch
  // left in supp_state / *d_nkeep; the state is stored back whenever supp_state is given.
  HIP_DYNAMIC_SHARED(unsigned char, smem);
  thr += blockIdx.z * bs.words; snap += blockIdx.z * bs.words; order += blockIdx.z * bs.order;
  keep_q += blockIdx.z * bs.ints; keep += blockIdx.z * bs.keep; d_nkeep += blockIdx.z;
  if (supp_state) supp_state += blockIdx.z * bs.ints / 2;   // per-frame workspaces are bs.ints 4-byte words apart
  unsigned long long* supp = (unsigned long long*)smem;      // [nwcap]
  unsigned long long* tile = supp + nwcap;                   // [64][tile_w]
  const int lane = threadIdx.x;
  const int K = d_count ? min(d_count[blockIdx.z], cap) : cap;
  const int nw = (K + 63) >> 6;
  for (int w = lane; w < nw; w += 64) supp[w] = c_begin > 0 ? supp_state[w] : 0ull;
  __builtin_amdgcn_wave_barrier();
  int M = c_begin > 0 ? *d_nkeep : 0;
  for (int c = c_begin; c < min(nw, c_end); ++c) {
    const int rows = min(64, K - (c << 6));
    // stage only the rows that are still unsuppressed when the chunk starts (the keep loop below never reads another
    // row, and in the second round most chunks have none): batches of up to 16 independent row loads in flight
    unsigned long long live = ~supp[c];
    if (rows < 64) live &= (1ull << rows) - 1ull;
    if (live == 0ull) continue;
    unsigned long long kept = 0ull;                          // rows of this chunk that the keep loop kept
    const int M0 = M;
    for (int wb = c; wb < nw; wb += tile_w) {                // column chunk [wb, we)
      const int we = min(nw, wb + tile_w);
      const unsigned long long need = wb == c ? live : kept;
      for (int w0 = wb; w0 < we; w0 += 64) {
        const int w = w0 + lane;
        unsigned long long rem = need;
        while (rem) {
          int rr[16];
          unsigned long long v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            rr[u] = rem ? __ffsll(rem) - 1 : -1;
            rem &= rem - 1ull;                               // (0 stays 0)
          }
#pragma unroll
          for (int u = 0; u < 16; ++u)
            v[u] = (w < we && rr[u] >= 0) ? thr[(size_t)((c << 6) + rr[u]) * nwcap + w] : 0ull;
#pragma unroll
          for (int u = 0; u < 16; ++u)
            if (w < we && rr[u] >= 0) tile[rr[u] * tile_w + (w - wb)] = v[u];
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (wb == c) {
        int r = 0;
        while (r < rows) {
          unsigned long long avail = ~supp[c] & (~0ull << r);
          if (rows < 64) avail &= (1ull << rows) - 1ull;
          if (avail == 0ull) break;
          r = __ffsll(avail) - 1;
          const int q = (c << 6) + r;
          for (int w = wb + lane; w < we; w += 64) {
            const unsigned long long sw = supp[w];
            snap[(size_t)M * nwcap + w] = sw;
            supp[w] = sw | tile[r * tile_w + (w - wb)];
          }
          if (lane == 0) {
            keep_q[M] = q;
            keep[M] = order[q];
          }
          kept |= 1ull << r;
          ++M;
          ++r;
          __builtin_amdgcn_wave_barrier();
        }
      } else {
        int i = 0;
        for (unsigned long long rem = kept; rem; rem &= rem - 1ull, ++i) {
          const int r = __ffsll(rem) - 1;
          for (int w = wb + lane; w < we; w += 64) {          // each lane only ever touches its own words: no barrier
            const unsigned long long sw = supp[w];
            snap[(size_t)(M0 + i) * nwcap + w] = sw;
            supp[w] = sw | tile[r * tile_w + (w - wb)];
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  if (lane == 0) *d_nkeep = M;
  if (supp_state)
    for (int w = lane; w < nw; w += 64) supp_state[w] = supp[w];
}

// Greedy scan, capacities up to 8 192 rows (the pipeline's): one workgroup of four wavefronts per frame.  The scan is a chain of
// dependent steps; what the single-wave form above spends its time on is memory latency -- one (or, with 64 live rows, four)
// round trips per 64-row chunk, 24 chunks at 1 500 candidates.  Here all four waves stage the rows of as many consecutive
// chunks as the LDS tile holds at the frame's ACTUAL width (nw - c words per row: five chunks at 1 500 candidates, so the first
// round is one round trip and the second four), every thread with 16 loads in flight; wave 0 then resolves those chunks out of
// LDS exactly as above (rows staged while still alive and suppressed later in the same group are simply never read).
__global__ __launch_bounds__(256) void wnms_scan4_kernel(const unsigned long long* __restrict__ thr,
                                                        unsigned long long* __restrict__ snap, int cap,
                                                        const int* __restrict__ d_count, int nwcap,
                                                        const int* __restrict__ order, int* __restrict__ keep_q,
                                                        int* __restrict__ keep, int* __restrict__ d_nkeep, WnmsBatch bs,
                                                        int c_begin, int c_end, unsigned long long* __restrict__ supp_state,
                                                        int tile_words) {
  HIP_DYNAMIC_SHARED(unsigned char, smem);
  thr += blockIdx.z * bs.words; snap += blockIdx.z * bs.words; order += blockIdx.z * bs.order;
  keep_q += blockIdx.z * bs.ints; keep += blockIdx.z * bs.keep; d_nkeep += blockIdx.z;
  if (supp_state) supp_state += blockIdx.z * bs.ints / 2;   // per-frame workspaces are bs.ints 4-byte words apart
  unsigned long long* supp = (unsigned long long*)smem;      // [nwcap]
  unsigned long long* tile = supp + nwcap;                   // [tile_words] >= 64 * nwcap
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = d_count ? min(d_count[blockIdx.z], cap) : cap;
  const int nw = (K + 63) >> 6;
  for (int w = tid; w < nw; w += 256) supp[w] = c_begin > 0 ? supp_state[w] : 0ull;
  __syncthreads();
  int M = c_begin > 0 ? *d_nkeep : 0;
  const int cend = min(nw, c_end);
  for (int c = c_begin; c < cend;) {
    const int P = nw - c;                                    // words per staged row: from the group's first diagonal word on
    const int ce = min(cend, c + max(1, tile_words / (64 * P)));
    const int nrows = min(K, ce << 6) - (c << 6);
    // stage the group's rows that are alive now
    for (int i0 = 0; i0 < nrows * P; i0 += 256 * 16) {
      unsigned long long v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int idx = i0 + u * 256 + tid;
        v[u] = 0ull;
        if (idx < nrows * P) {
          const int r = idx / P, w = idx - r * P;
          if (!((supp[c + (r >> 6)] >> (r & 63)) & 1ull)) v[u] = thr[(size_t)((c << 6) + r) * nwcap + c + w];
        }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int idx = i0 + u * 256 + tid;
        if (idx < nrows * P) tile[idx] = v[u];
      }
    }
    __syncthreads();
    if (wave == 0) {
      for (int cc = c; cc < ce; ++cc) {
        const int rows = min(64, K - (cc << 6));
        int r = 0;
        while (r < rows) {
          unsigned long long avail = ~supp[cc] & (~0ull << r);
          if (rows < 64) avail &= (1ull << rows) - 1ull;
          if (avail == 0ull) break;
          r = __ffsll(avail) - 1;
          const unsigned long long* trow = tile + (size_t)(((cc - c) << 6) + r) * P - c;   // trow[w] = thr word w of this row
          for (int w = cc + lane; w < nw; w += 64) {
            const unsigned long long sw = supp[w];
            snap[(size_t)M * nwcap + w] = sw;
            supp[w] = sw | trow[w];
          }
          if (lane == 0) {
            keep_q[M] = (cc << 6) + r;
            keep[M] = order[(cc << 6) + r];
          }
          ++M;
          ++r;
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    __syncthreads();
    c = ce;
  }
  if (tid == 0) *d_nkeep = M;
  if (supp_state)
    for (int w = tid; w < nw; w += 256) supp_state[w] = supp[w];
}

// Rows >= first_row that the first round left unsuppressed, ascending -> rows_out, their number -> *nrows_out.
// One workgroup of 256 threads per frame; suppression words are taken 256 at a time with a running output offset.
__global__ __launch_bounds__(256) void wnms_alive_kernel(const unsigned long long* __restrict__ supp_state, int cap,
                                                         const int* __restrict__ d_count, int first_row,
                                                         int* __restrict__ rows_out, int* __restrict__ nrows_out, WnmsBatch bs) {
  supp_state += blockIdx.z * bs.ints / 2; rows_out += blockIdx.z * bs.ints; nrows_out += blockIdx.z * bs.ints;
  const int K = d_count ? min(d_count[blockIdx.z], cap) : cap;
  const int nw = (K + 63) >> 6, tid = threadIdx.x;
  __shared__ int part[256];
  int base = 0;
  for (int w0 = 0; w0 < nw; w0 += 256) {
    const int w = w0 + tid;
    unsigned long long alive = 0ull;
    if (w < nw) {
      alive = ~supp_state[w];
      const int lo = first_row - (w << 6), hi = K - (w << 6);             // keep rows in [first_row, K)
      if (lo >= 64) alive = 0ull; else if (lo > 0) alive &= ~0ull << lo;
      if (hi < 64) alive &= hi <= 0 ? 0ull : (1ull << hi) - 1ull;
    }
    const int cnt = __popcll(alive);
    part[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const int v = tid >= off ? part[tid - off] : 0;
      __syncthreads();
      part[tid] += v;
      __syncthreads();
    }
    int pos = base + part[tid] - cnt;
    while (alive) {
      const int b = __ffsll(alive) - 1;
      rows_out[pos++] = (w << 6) + b;
      alive &= alive - 1ull;
    }
    base += part[255];
    __syncthreads();
  }
  if (tid == 0) *nrows_out = base;
}

// One kept row: neighbourhood list, median yaw, weighted sums.  nbl / yws hold n + 1 entries (LDS in the common case, a
// per-frame global scratch for the rare row whose neighbourhood does not fit -- see wnms_merge_big_kernel).
__device__ void wnms_merge_row(const float* __restrict__ dets, const int* __restrict__ order,
                               const unsigned long long* __restrict__ vrow, const unsigned long long* __restrict__ srow,
                               int nw, int q, int* nbl, float* yws, int lane, float* __restrict__ orow, bool global_scratch) {
  RD_NOCONTRACT
  const int irow = order[q];
  const float yaw_i = dets[(size_t)irow * 12 + 8];
  // neighbourhood list in ascending sorted position, self first (nms.h:496-517)
  int n = 1;
  if (lane == 0) {
    nbl[0] = q;
    yws[0] = yaw_i;
  }
  for (int w = q >> 6; w < nw; ++w) {
    unsigned long long word = vrow[w] & ~srow[w];
    if ((word >> lane) & 1ull) {
      int pos = n + __popcll(word & ((1ull << lane) - 1ull));
      int q2 = (w << 6) + lane;
      nbl[pos] = q2;
      yws[pos] = dets[(size_t)order[q2] * 12 + 8];
    }
    n += __popcll(word);
  }
  if (global_scratch) __threadfence();
  __syncthreads();
  // median yaw (nms.h:527-540)
  float med = yaw_i;
  if (n > 2) {
    int sz = n;
    if ((n & 1) == 0) {
      if (lane == 0) yws[n] = yaw_i;
      sz = n + 1;
    }
    if (global_scratch) __threadfence();
    __syncthreads();
    const int target = sz >> 1;
    float found = 0.f;
    int have = 0;
    for (int c0 = 0; c0 < sz; c0 += 64) {
      int c = c0 + lane;
      int ok = 0;
      float v = 0.f;
      if (c < sz) {
        v = yws[c];
        int less = 0, eq = 0;
        for (int k = 0; k < sz; ++k) {
          float u = yws[k];
          less += (u < v);
          eq += (u == v);
        }
        ok = (less <= target) && (target < less + eq);
      }
      unsigned long long bm = __ballot(ok);
      if (bm != 0ull && !have) {
        int src = __ffsll(bm) - 1;
        found = __shfl(v, src);
        have = 1;
      } else {
        (void)__shfl(v, 0);  // keep the collective wave-uniform
      }
    }
    if (have) med = found;
  }
  // weighted sums in neighbourhood order (nms.h:541-573): lane f < 11 owns field f
  float sum1 = 0.f, sum3 = 0.f;
  for (int k = 0; k < n; ++k) {
    const float* r = dets + (size_t)order[nbl[k]] * 12;
    float yl = r[8];
    if ((double)fmodf(fabsf(yl - med), float(2 * 3.1415926)) >= 0.3) continue;
    float p = r[11];
    if (lane < 11) {
      sum1 += p * r[lane];
      sum3 += p;
    }
  }
  if (lane < 11) orow[lane] = sum1 / sum3;
  if (lane == 11) orow[11] = dets[(size_t)irow * 12 + 11];
}

// One wavefront per kept row.  lds_cap = entries of the LDS neighbourhood list; a row whose neighbourhood (+ the median's
// extra slot) does not fit is appended to the frame's overflow list and handled by wnms_merge_big_kernel.
__global__ __launch_bounds__(64) void wnms_merge_kernel(const float* __restrict__ dets, const int* __restrict__ order,
                                                        const unsigned long long* __restrict__ vote,
                                                        const unsigned long long* __restrict__ snap, int cap,
                                                        const int* __restrict__ d_count, int nwcap,
                                                        const int* __restrict__ keep_q, const int* __restrict__ d_nkeep,
                                                        float* __restrict__ out, WnmsBatch bs, int lds_cap,
                                                        int* __restrict__ ovf, int* __restrict__ novf) {
  dets += blockIdx.z * bs.dets; order += blockIdx.z * bs.order; vote += blockIdx.z * bs.words; snap += blockIdx.z * bs.words;
  keep_q += blockIdx.z * bs.ints; out += blockIdx.z * bs.out; ovf += blockIdx.z * bs.ints; novf += blockIdx.z * bs.ints;
  HIP_DYNAMIC_SHARED(unsigned char, smem);  // lds_cap ints + lds_cap floats
  const int lane = threadIdx.x;
  const int K = d_count ? min(d_count[blockIdx.z], cap) : cap;
  const int nw = (K + 63) >> 6;
  const int nkeep = d_nkeep[blockIdx.z];
  for (int mrow = blockIdx.x; mrow < nkeep; mrow += gridDim.x) {
    const int q = keep_q[mrow];
    const unsigned long long* vrow = vote + (size_t)q * nwcap;
    const unsigned long long* srow = snap + (size_t)mrow * nwcap;
    if (K + 2 > lds_cap) {   // the list might not fit: count first
      int n = 1;
      for (int w = (q >> 6) + lane; w < nw; w += 64) n += __popcll(vrow[w] & ~srow[w]);
      for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off);
      n -= 63;                                               // every lane started from 1
      if (n + 1 > lds_cap) {
        if (lane == 0) ovf[atomicAdd(novf, 1)] = mrow;
        continue;
      }
    }
    __syncthreads();                                         // previous row's list readers are done
    wnms_merge_row(dets, order, vrow, srow, nw, q, (int*)smem, (float*)(smem + (size_t)lds_cap * 4), lane,
                   out + (size_t)mrow * 12, false);
  }
}
// The overflow rows of a frame, one after the other, with the list in the frame's global scratch (cap + 2 entries).
__global__ __launch_bounds__(64) void wnms_merge_big_kernel(const float* __restrict__ dets, const int* __restrict__ order,
                                                            const unsigned long long* __restrict__ vote,
                                                            const unsigned long long* __restrict__ snap, int cap,
                                                            const int* __restrict__ d_count, int nwcap,
                                                            const int* __restrict__ keep_q, float* __restrict__ out,
                                                            WnmsBatch bs, const int* __restrict__ ovf,
                                                            const int* __restrict__ novf, int* __restrict__ scratch) {
  dets += blockIdx.z * bs.dets; order += blockIdx.z * bs.order; vote += blockIdx.z * bs.words; snap += blockIdx.z * bs.words;
  keep_q += blockIdx.z * bs.ints; out += blockIdx.z * bs.out; ovf += blockIdx.z * bs.ints; novf += blockIdx.z * bs.ints;
  scratch += blockIdx.z * bs.ints;
  const int n_ovf = *novf;
  if (n_ovf == 0) return;
  const int K = d_count ? min(d_count[blockIdx.z], cap) : cap;
  const int nw = (K + 63) >> 6;
  for (int i = 0; i < n_ovf; ++i) {
    const int mrow = ovf[i], q = keep_q[mrow];
    __syncthreads();
    wnms_merge_row(dets, order, vote + (size_t)q * nwcap, snap + (size_t)mrow * nwcap, nw, q, scratch,
                   (float*)(scratch + cap + 2), threadIdx.x, out + (size_t)mrow * 12, true);
    __threadfence();
  }
}

// ---- processing order of the reference: std::sort with an unstable comparator on ties (nms.h:786-792) ----------------------
// point4_wnms_4c sorts iota(K) with `score[i] > score[j]` through libstdc++'s std::sort, whose result for tied scores is
// whatever introsort leaves.  This kernel replays that algorithm (GCC libstdc++ bits/stl_algo.h: __introsort_loop with
// median-of-three to first + unguarded Hoare partition, threshold 16, depth limit 2*floor(log2 n) -> heap sort, then the
// final insertion sort) decision by decision, one wavefront per frame: the partition scans advance 64 elements per step
// (ballot), the sub-ranges are independent of the order in which they are processed (explicit stack), and the final
// insertion sort never moves an element across the boundary of a finished range (left >= pivot >= right), so the
// finished ranges are insertion-sorted independently, one lane per range.  Strictly decreasing input (the common case in
// the pipeline: rows arrive sorted, ties are rare) is recognised up front: every correct sort returns the identity.
struct TieSort {
  float* sc; int* ix; unsigned* segbit; unsigned* tmbit; unsigned* idbit; int n, lane; bool fence;
  __device__ __forceinline__ void sync() const {
    if (fence) __threadfence();
    __builtin_amdgcn_wave_barrier();
  }
  // (the leading barrier orders "every lane has read its operands" before lane 0 writes: free on hardware, where the
  // wave runs in lockstep; required under hipemu, where lanes are fibers)
  __device__ __forceinline__ void swap(int a, int b) const {
    __builtin_amdgcn_wave_barrier();
    if (lane == 0 && a != b) {
      const float t = sc[a]; sc[a] = sc[b]; sc[b] = t;
      const int u = ix[a]; ix[a] = ix[b]; ix[b] = u;
    }
    sync();
  }
  __device__ __forceinline__ void mark(int first) const {
    if (lane == 0) segbit[first >> 5] |= 1u << (first & 31);
  }
  __device__ int partition_pivot(int first, int last) const {          // __unguarded_partition_pivot
    const int mid = first + (last - first) / 2;
    const int a = first + 1, b = mid, c = last - 1;                    // __move_median_to_first(first, a, b, c)
    const float sa = sc[a], sb = sc[b], s3 = sc[c];
    if (sa > sb) {
      if (sb > s3) swap(first, b); else if (sa > s3) swap(first, c); else swap(first, a);
    } else if (sa > s3) swap(first, a);
    else if (sb > s3) swap(first, c);
    else swap(first, b);
    const float pv = sc[first];
    int f = first + 1, l = last;                                       // __unguarded_partition(first + 1, last, first)
    for (;;) {
      for (;;) {                                                       // while (comp(f, pivot)) ++f;
        const int p = f + lane;
        const bool go = p < n && sc[p] > pv;
        const unsigned long long stop = __ballot(!go);
        if (stop) { f += __ffsll(stop) - 1; break; }
        f += 64;
      }
      --l;
      for (;;) {                                                       // while (comp(pivot, l)) --l;
        const int p = l - lane;
        const bool go = p >= 0 && pv > sc[p];
        const unsigned long long stop = __ballot(!go);
        if (stop) { l -= __ffsll(stop) - 1; break; }
        l -= 64;
      }
      if (!(f < l)) return f;
      swap(f, l);
      ++f;
    }
  }
  // heap sort of [first, last): __partial_sort(first, last, last) = __make_heap + __sort_heap (all lanes in lockstep, lane 0 writes)
  __device__ void put(int p, float s, int i) const {
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) { sc[p] = s; ix[p] = i; }
    sync();
  }
  __device__ void adjust_heap(int first, int hole, int len, float vs, int vi) const {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      if (sc[first + child] > sc[first + child - 1]) --child;
      put(first + hole, sc[first + child], ix[first + child]);
      hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
      child = 2 * (child + 1);
      put(first + hole, sc[first + child - 1], ix[first + child - 1]);
      hole = child - 1;
    }
    int parent = (hole - 1) / 2;                                       // __push_heap
    while (hole > top && sc[first + parent] > vs) {
      put(first + hole, sc[first + parent], ix[first + parent]);
      hole = parent;
      parent = (hole - 1) / 2;
    }
    put(first + hole, vs, vi);
  }
  __device__ void heap_sort(int first, int last) const {
    const int len = last - first;
    if (len >= 2)
      for (int parent = (len - 2) / 2;; --parent) {                    // __make_heap
        adjust_heap(first, parent, len, sc[first + parent], ix[first + parent]);
        if (parent == 0) break;
      }
    for (int l = last; l - first > 1;) {                               // __sort_heap / __pop_heap
      --l;
      const float vs = sc[l]; const int vi = ix[l];
      put(l, sc[first], ix[first]);
      adjust_heap(first, 0, l - first, vs, vi);
    }
  }
  // SORTED INPUT (the pipeline's case: rows arrive in score order, ties are few).  A range that holds no member of a tie
  // group ends, whatever introsort does inside it, as the identity: a correct sort puts an element without a tie partner
  // at its rank, and in sorted input rank == index.  Such ranges are not replayed at all (idbit: filled with p -> p at the
  // end); only ranges that contain tied rows are -- a few dozen partitions instead of K / 12.
  __device__ bool has_tie(int f, int l) const {
    for (int p0 = f; p0 < l; p0 += 64) {
      const int p = p0 + lane;
      const int i = p < l ? ix[p] : 0;
      const bool t = p < l && ((tmbit[i >> 5] >> (i & 31)) & 1u);
      if (__ballot(t)) return true;
    }
    return false;
  }
  __device__ void run(int* stack, bool sorted_input) const {
    if (n < 2) return;
    int sp = 0, first = 0, last = n, depth = 2 * (31 - __clz(n));
    for (;;) {
      bool skipped = sorted_input && !has_tie(first, last);
      while (!skipped && last - first > 16) {                          // __introsort_loop
        if (depth == 0) { heap_sort(first, last); break; }
        --depth;
        const int cut = partition_pivot(first, last);
        stack[3 * sp] = cut; stack[3 * sp + 1] = last; stack[3 * sp + 2] = depth; ++sp;
        last = cut;
        skipped = sorted_input && !has_tie(first, last);
      }
      if (first < n && lane == 0) {
        segbit[first >> 5] |= 1u << (first & 31);
        if (skipped) idbit[first >> 5] |= 1u << (first & 31);
      }
      if (sp == 0) break;
      --sp;
      first = stack[3 * sp]; last = stack[3 * sp + 1]; depth = stack[3 * sp + 2];
    }
    sync();
    // __final_insertion_sort, range by range: lane owns the ranges that start in its bitmap words
    const int nwords = (n + 31) >> 5;
    for (int w = lane; w < nwords; w += 64) {
      unsigned bits = segbit[w];
      const unsigned idb = idbit[w];
      while (bits) {
        const int b0 = __ffs(bits) - 1;
        const int s = (w << 5) + b0;
        bits &= bits - 1u;
        int e = n;                                                      // next range start
        {
          unsigned rest = bits;
          int ww = w;
          while (!rest && ++ww < nwords) rest = segbit[ww];
          if (rest) e = (ww << 5) + __ffs(rest) - 1;
        }
        if ((idb >> b0) & 1u) {
          for (int i = s; i < e; ++i) ix[i] = i;
          continue;
        }
        for (int i = s + 1; i < e; ++i) {
          const float v = sc[i]; const int vi = ix[i];
          int j = i;
          while (j > s && v > sc[j - 1]) { sc[j] = sc[j - 1]; ix[j] = ix[j - 1]; --j; }
          sc[j] = v; ix[j] = vi;
        }
      }
    }
    sync();
  }
};
constexpr int TIE_STACK_BYTES = 3 * 40 * 4 + 16;   // introsort range stack + the two detection flags in front of it
constexpr int TIE_LDS_K = 16384;   // keys + indices + range bitmap in LDS up to this K (130 KB); beyond it: global scratch
// 256 threads scan the scores (identity fill + "any tie / any rise"): that pass is all there is to do for most frames, and as a
// single wave it kept the kernel resident for ~100 us (128 dependent iterations); wave 0 alone runs the replay when needed.
__global__ __launch_bounds__(256) void wnms_tie_order_kernel(const float* __restrict__ dets, int cap,
                                                            const int* __restrict__ d_count, int* __restrict__ order,
                                                            WnmsBatch bs, long order_stride, int* __restrict__ scratch) {
  HIP_DYNAMIC_SHARED(unsigned char, smem);
  dets += blockIdx.z * bs.dets; order += blockIdx.z * order_stride; scratch += blockIdx.z * bs.ints;
  const int tid = threadIdx.x, lane = tid & 63;
  const int K = d_count ? min(d_count[blockIdx.z], cap) : cap;
  int* flags = (int*)smem;                       // [0] any tie, [1] any rise
  if (tid < 2) flags[tid] = 0;
  __syncthreads();
  // identity for the unused tail; strictly decreasing scores -> identity everywhere; non-increasing (sorted with ties) ->
  // only the ranges holding tied rows are replayed
  int tied = 0, rising = 0;
  for (int i = tid; i < cap; i += 256) {
    order[i] = i;
    if (i + 1 < K) {
      const float a0 = dets[(size_t)i * 12 + 11], a1 = dets[(size_t)(i + 1) * 12 + 11];
      if (a0 == a1) tied = 1;
      else if (!(a0 > a1)) rising = 1;
    }
  }
  if (tied) flags[0] = 1;
  if (rising) flags[1] = 1;
  __threadfence();                               // the identity fill is in memory before wave 0 may overwrite parts of it
  __syncthreads();
  const bool any_tied = flags[0] != 0, any_rising = flags[1] != 0;
  if ((!any_tied && !any_rising) || tid >= 64) return;   // (whole waves leave: the replay only uses wave-level barriers)
  int* stack = (int*)(smem + 16);                // [3 * 40] introsort range stack, then (LDS variant) keys / indices / bitmaps
  unsigned char* lds = smem + TIE_STACK_BYTES;
  TieSort T;
  T.n = K; T.lane = lane;
  const int nbit = (K + 31) >> 5, nbcap = cap / 32 + 2;
  if (cap <= TIE_LDS_K) {
    T.sc = (float*)lds; T.ix = (int*)(lds + (size_t)cap * 4); T.segbit = (unsigned*)(lds + (size_t)cap * 8); T.fence = false;
  } else {
    T.sc = (float*)scratch; T.ix = scratch + cap; T.segbit = (unsigned*)(scratch + 2 * (size_t)cap); T.fence = true;
  }
  T.tmbit = T.segbit + nbcap; T.idbit = T.tmbit + nbcap;
  for (int i = lane; i < K; i += 64) { T.sc[i] = dets[(size_t)i * 12 + 11]; T.ix[i] = i; }
  for (int i = lane; i < nbit; i += 64) { T.segbit[i] = 0u; T.tmbit[i] = 0u; T.idbit[i] = 0u; }
  T.sync();
  if (!any_rising) {   // tie membership by original index: a row whose score equals a neighbour's
    for (int w = lane; w < nbit; w += 64) {
      unsigned bits = 0u;
      for (int b = 0; b < 32; ++b) {
        const int i = (w << 5) + b;
        if (i < K && ((i > 0 && T.sc[i] == T.sc[i - 1]) || (i + 1 < K && T.sc[i] == T.sc[i + 1]))) bits |= 1u << b;
      }
      T.tmbit[w] = bits;
    }
    T.sync();
  }
  T.run(stack, !any_rising);
  for (int i = lane; i < K; i += 64) order[i] = T.ix[i];
}

// OverlapChecker::single_overlap (nms.h:195-249) for n independent row pairs: out[i] = overlap(a[i], b[i]) with a as the
// first (kept) box.  Rows are (.., 12) dets rows (8 corners, yaw, bottom, height, score); the per-box preparation is the
// same code as wnms_prep_kernel's (normalised winding, edge angles, area).
__device__ __forceinline__ void w_prep_box(const float* b, float* o) {
  RD_NOCONTRACT
  WPt p[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) p[k] = {b[2 * k], b[2 * k + 1]};
  bool cw = ((p[1].x - p[0].x) * (p[2].y - p[0].y) - (p[2].x - p[0].x) * (p[1].y - p[0].y)) > 0;
  if (cw) {
    WPt t0 = p[0], t1 = p[1];
    p[0] = p[3]; p[1] = p[2]; p[2] = t1; p[3] = t0;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[2 * k] = p[k].x;
    o[2 * k + 1] = p[k].y;
    int k1 = (k + 1) & 3;
    o[8 + k] = fdlibm_atan2f(p[k1].y - p[k].y, p[k1].x - p[k].x);
  }
  float area = 0.f;
  area += w_cross3(p[0], p[1], p[2]);
  area += w_cross3(p[0], p[2], p[3]);
  if (area < 0) area = -area;
  o[12] = area / 2;
  o[13] = b[9];
  o[14] = b[10];
  o[15] = 0.f;
}
__global__ __launch_bounds__(64) void single_overlap_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                            int is3d, float* __restrict__ out) {
  __shared__ float edges[EDGE_LDS_BYTES / 4];
  __shared__ float pb[64 * 16];
  const int t = threadIdx.x;
  EdgeLds EL;
  EL.ax = edges; EL.ay = edges + 512; EL.bx = edges + 1024; EL.by = edges + 1536; EL.an = edges + 2048;
  EL.t = t;
  const long i = (long)blockIdx.x * 64 + t;
  if (i >= n) return;
  float pa[16];
  w_prep_box(a + i * 12, pa);
  w_prep_box(b + i * 12, pb + t * 16);
  out[i] = w_overlap(pa, pb + t * 16, is3d != 0, EL);
}

// the rejection test of the pair kernel on n independent row pairs (tests / characterisation only): out[i] = 1 if the pair kernel
// would not clip (a[i], b[i])
__global__ __launch_bounds__(64) void pair_skippable_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                            unsigned char* __restrict__ out) {
  const long i = (long)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  float pa[PREP_F], pb[PREP_F];
  w_box_domain(a + i * 12, pa);
  w_box_domain(b + i * 12, pb);
  out[i] = w_pair_skippable(pa, pb) ? 1 : 0;
}

__global__ __launch_bounds__(256) void iota_order_kernel(int* order, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) order[i] = i;
}

struct WnmsWs {
  float* prep;
  unsigned long long *thr, *vote, *snap, *supp_state;
  int *keep_q, *order, *alive, *nalive, *ovf, *novf, *scratch;
  void* sort_ws;
  int nwcap;
};
inline size_t wnms_ws_bytes(int cap);
inline WnmsWs wnms_ws_carve(void* ws, int cap);

}  // namespace rd
