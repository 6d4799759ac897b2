// Weighted NMS (operator_cxx/src_cxx/nms.h:452-577 in the reference), decomposed for the GPU so that every
// float operation happens in the reference's order (this file is compiled with FP contraction OFF):
//   1. prep   : per box (in processing order) clockwise-normalised corners, the 4 edge polar angles (the only
//               atan2f of the algorithm -- hoisted out of the O(K^2) part), area, bottom, height.
//   2. pairs  : for sorted positions q1 < q2 the half-plane-intersection IoU (nms.h:96-249) -> two K x K/64
//               bit matrices  thr[q1] (ovr >= thresh)  and  vote[q1] (ovr > thresh_vote).  No spatial prefilter:
//               the reference's 100 m hash passes every pair inside +-100 m, and an AABB reject is NOT
//               result-neutral (disjoint boxes can produce a non-zero "IoU" in the reference's clipper).
//   3. scan   : one wavefront walks the sorted positions; for an unsuppressed q it snapshots
//               nb[q] = vote[q] & ~supp (written back in place) and ORs thr[q] into supp (64 words per step).
//   4. merge  : one wavefront per kept row: median yaw by rank selection, then the 11 score-weighted sums in
//               neighbourhood order (lane f accumulates field f sequentially -> same rounding as the reference).
#pragma once
#include "rd_common.h"

namespace rd {

struct WPt { float x, y; };
struct WEdge { WPt a, b; float ang; };
constexpr int PREP_F = 16;  // floats per prepped box: 8 corners, 4 angles, area, bottom, height, pad

#define RD_NOCONTRACT _Pragma("clang fp contract(off)")

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
  int* dq;                        // [16][64]
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
  __device__ __forceinline__ int& q(int i) const { return dq[i * 64 + t]; }
};
constexpr int EDGE_LDS_BYTES = (5 * 8 + 16) * 64 * 4;

// box1 = the earlier (kept candidate) box, box2 = the later one; pre1/pre2 their prepped records.
__device__ float w_overlap(const float* pre1, const float* pre2, bool is3d, const EdgeLds& L) {
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
  float area1 = pre1[12], area2 = pre2[12];
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
  const int t = j + 1;
  L.q(0) = 0;
  L.q(1) = 1;
  int top = 1, bot = 0;
  for (i = 2; i < t; i++) {
    const WEdge li = L.get(i);
    while (top > bot && w_outside(li, L.get(L.q(top)), L.get(L.q(top - 1)))) top--;
    while (top > bot && w_outside(li, L.get(L.q(bot)), L.get(L.q(bot + 1)))) bot++;
    L.q(++top) = i;
  }
  while (top > bot && w_outside(L.get(L.q(bot)), L.get(L.q(top)), L.get(L.q(top - 1)))) top--;
  while (top > bot && w_outside(L.get(L.q(top)), L.get(L.q(bot)), L.get(L.q(bot + 1)))) bot++;
  {
    const int qb = L.q(bot);
    L.q(++top) = qb;
  }
  // polygon vertices + fan area (nms.h:147-166), vertices generated on the fly
  const int nv = top - bot;
  float inter = 0.f;
  if (nv >= 3) {
    WPt p0 = w_meet(L.get(L.q(bot + 1)), L.get(L.q(bot)));
    WPt pa = w_meet(L.get(L.q(bot + 2)), L.get(L.q(bot + 1)));
    float area = 0.f;
    for (int k = 2; k < nv; ++k) {
      WPt pb = w_meet(L.get(L.q(bot + k + 1)), L.get(L.q(bot + k)));
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
                                                        WnmsBatch bs) {
  RD_NOCONTRACT
  dets += blockIdx.z * bs.dets; order += blockIdx.z * bs.order; prep += blockIdx.z * bs.prep;
  int q = blockIdx.x * 256 + threadIdx.x;
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
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[2 * k] = p[k].x;
    o[2 * k + 1] = p[k].y;
    int k1 = (k + 1) & 3;
    o[8 + k] = atan2f(p[k1].y - p[k].y, p[k1].x - p[k].x);  // nms.h:71
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

// grid (8*nb, nb), 64 threads: rows 64*blockIdx.y.. x the 8 columns 64*cb + 8*sub.. (cb = blockIdx.x / 8, upper
// triangle only).  One wave evaluates 64 x 8 pairs and writes ONE BYTE of each row's 64-bit mask word (little endian:
// byte `sub` holds bits 8*sub..8*sub+7), so K = 1.5k boxes already give ~2.4k waves for the 256 CUs.
constexpr int WN_CT = 8;
__global__ __launch_bounds__(64) void wnms_pairs_kernel(const float* __restrict__ prep, int cap,
                                                        const int* __restrict__ d_count, float thresh, float thresh_vote,
                                                        int is3d, unsigned long long* __restrict__ thr,
                                                        unsigned long long* __restrict__ vote, int nwcap, WnmsBatch bs,
                                                        const int* __restrict__ rows, const int* __restrict__ nrows,
                                                        const unsigned long long* __restrict__ supp_state) {
  // rows == nullptr: row block blockIdx.y is rows 64*blockIdx.y .. (first round).  Otherwise (second round) the row
  // block is 64 consecutive entries of the compacted, ascending list of rows that survived the first round.
  prep += blockIdx.z * bs.prep; thr += blockIdx.z * bs.words; vote += blockIdx.z * bs.words;
  const int rb = blockIdx.y, cb = blockIdx.x / WN_CT, sub = blockIdx.x % WN_CT;
  const int K = d_count ? min(d_count[blockIdx.z], cap) : cap;
  int nr = K, qmin = rb * 64;
  unsigned dead = 0u;   // second round: columns of this tile that the first round already suppressed.  Neither their thr
                        // bit (ORed into a suppression word that already has it) nor their vote bit (masked by the
                        // suppression snapshot in the merge) can influence anything, so those pairs are not evaluated.
  if (rows) {
    dead = (unsigned)(supp_state[blockIdx.z * (bs.ints / 2) + cb] >> (sub * WN_CT)) & 0xffu;
    rows += blockIdx.z * bs.ints;
    nr = nrows[blockIdx.z * bs.ints];                      // (per-frame workspaces are bs.ints 4-byte words apart)
    if (rb * 64 >= nr) return;
    qmin = rows[rb * 64];
  }
  // skip column tiles that lie entirely before the first row (the tile holding that row itself is still written: the
  // scan and the merge read every row's words from its own, diagonal, word on)
  if (cb * 64 + 63 < qmin || rb * 64 >= nr || cb * 64 >= K) return;
  __shared__ float colp[WN_CT * PREP_F];
  __shared__ float edges[EDGE_LDS_BYTES / 4];
  const int t = threadIdx.x;
  EdgeLds EL;
  EL.ax = edges; EL.ay = edges + 512; EL.bx = edges + 1024; EL.by = edges + 1536; EL.an = edges + 2048;
  EL.dq = (int*)(edges + 2560);
  EL.t = t;
  const int c0 = cb * 64 + sub * WN_CT;
  for (int i = t; i < WN_CT * PREP_F; i += 64) {
    int q2 = c0 + i / PREP_F;
    colp[i] = q2 < K ? prep[(size_t)q2 * PREP_F + (i % PREP_F)] : 0.f;
  }
  __syncthreads();
  if (rb * 64 + t >= nr) return;
  const int q1 = rows ? rows[rb * 64 + t] : rb * 64 + t;
  unsigned mt = 0u, mv = 0u;
  if (c0 + WN_CT - 1 > q1) {
    float mine[PREP_F];
#pragma unroll
    for (int k = 0; k < PREP_F; ++k) mine[k] = prep[(size_t)q1 * PREP_F + k];
    for (int c = 0; c < WN_CT; ++c) {
      int q2 = c0 + c;
      if (q2 < K && q2 > q1 && !((dead >> c) & 1u)) {
        float ovr = w_overlap(mine, &colp[c * PREP_F], is3d != 0, EL);
        if (ovr >= thresh) mt |= 1u << c;
        if (ovr > thresh_vote) mv |= 1u << c;
      }
    }
  }
  ((unsigned char*)thr)[((size_t)q1 * nwcap + cb) * 8 + sub] = (unsigned char)mt;
  ((unsigned char*)vote)[((size_t)q1 * nwcap + cb) * 8 + sub] = (unsigned char)mv;
}

// Greedy scan, one wavefront.  Per 64 sorted rows the needed part of the thr bit-matrix (64 rows x remaining words) is
// pulled into LDS with all loads in flight at once (one memory latency per 64 rows instead of a dependent load chain per
// kept row); the rows are then resolved sequentially out of LDS.  For every kept row the suppression state BEFORE it is
// snapshotted to `snap` (the merge kernel forms the neighbourhood vote[q] & ~snap), so the vote matrix is never read here.
__global__ __launch_bounds__(64) void wnms_scan_kernel(const unsigned long long* __restrict__ thr,
                                                       unsigned long long* __restrict__ snap, int cap,
                                                       const int* __restrict__ d_count, int nwcap,
                                                       const int* __restrict__ order, int* __restrict__ keep_q,
                                                       int* __restrict__ keep, int* __restrict__ d_nkeep, WnmsBatch bs,
                                                       int c_begin, int c_end, unsigned long long* __restrict__ supp_state) {
  // 64-row chunks [c_begin, c_end) only.  c_begin > 0 resumes from the suppression state / keep count a previous launch
  // left in supp_state / *d_nkeep; the state is stored back whenever supp_state is given.
  HIP_DYNAMIC_SHARED(unsigned char, smem);
  thr += blockIdx.z * bs.words; snap += blockIdx.z * bs.words; order += blockIdx.z * bs.order;
  keep_q += blockIdx.z * bs.ints; keep += blockIdx.z * bs.keep; d_nkeep += blockIdx.z;
  if (supp_state) supp_state += blockIdx.z * bs.ints / 2;   // per-frame workspaces are bs.ints 4-byte words apart
  unsigned long long* supp = (unsigned long long*)smem;      // [nwcap]
  unsigned long long* tile = supp + nwcap;                   // [64][nwcap]
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
    for (int w0 = c; w0 < nw; w0 += 64) {
      const int w = w0 + lane;
      unsigned long long rem = live;
      while (rem) {
        int rr[16];
        unsigned long long v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          rr[u] = rem ? __ffsll(rem) - 1 : -1;
          rem &= rem - 1ull;                                   // (0 stays 0)
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
          v[u] = (w < nw && rr[u] >= 0) ? thr[(size_t)((c << 6) + rr[u]) * nwcap + w] : 0ull;
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (w < nw && rr[u] >= 0) tile[rr[u] * nwcap + w] = v[u];
      }
    }
    __builtin_amdgcn_wave_barrier();
    int r = 0;
    while (r < rows) {
      unsigned long long avail = ~supp[c] & (~0ull << r);
      if (rows < 64) avail &= (1ull << rows) - 1ull;
      if (avail == 0ull) break;
      r = __ffsll(avail) - 1;
      const int q = (c << 6) + r;
      for (int w = c + lane; w < nw; w += 64) {
        const unsigned long long sw = supp[w];
        snap[(size_t)M * nwcap + w] = sw;
        supp[w] = sw | tile[r * nwcap + w];
      }
      if (lane == 0) {
        keep_q[M] = q;
        keep[M] = order[q];
      }
      ++M;
      ++r;
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (lane == 0) *d_nkeep = M;
  if (supp_state)
    for (int w = lane; w < nw; w += 64) supp_state[w] = supp[w];
}

// Rows >= first_row that the first round left unsuppressed, ascending -> rows_out, their number -> *nrows_out.
// One workgroup of 256 threads per frame, thread w owns suppression word w (K <= 16384 rows = 256 words).
__global__ __launch_bounds__(256) void wnms_alive_kernel(const unsigned long long* __restrict__ supp_state, int cap,
                                                         const int* __restrict__ d_count, int first_row,
                                                         int* __restrict__ rows_out, int* __restrict__ nrows_out, WnmsBatch bs) {
  supp_state += blockIdx.z * bs.ints / 2; rows_out += blockIdx.z * bs.ints; nrows_out += blockIdx.z * bs.ints;
  const int K = d_count ? min(d_count[blockIdx.z], cap) : cap;
  const int nw = (K + 63) >> 6, w = threadIdx.x;
  unsigned long long alive = 0ull;
  if (w < nw) {
    alive = ~supp_state[w];
    const int lo = first_row - (w << 6), hi = K - (w << 6);             // keep rows in [first_row, K)
    if (lo >= 64) alive = 0ull; else if (lo > 0) alive &= ~0ull << lo;
    if (hi < 64) alive &= hi <= 0 ? 0ull : (1ull << hi) - 1ull;
  }
  __shared__ int part[256];
  const int cnt = __popcll(alive);
  part[w] = cnt;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int v = w >= off ? part[w - off] : 0;
    __syncthreads();
    part[w] += v;
    __syncthreads();
  }
  int pos = part[w] - cnt;
  while (alive) {
    const int b = __ffsll(alive) - 1;
    rows_out[pos++] = (w << 6) + b;
    alive &= alive - 1ull;
  }
  if (w == 255) *nrows_out = part[255];
}

__global__ __launch_bounds__(64) void wnms_merge_kernel(const float* __restrict__ dets, const int* __restrict__ order,
                                                        const unsigned long long* __restrict__ vote,
                                                        const unsigned long long* __restrict__ snap, int cap,
                                                        const int* __restrict__ d_count, int nwcap,
                                                        const int* __restrict__ keep_q, const int* __restrict__ d_nkeep,
                                                        float* __restrict__ out, WnmsBatch bs) {
  RD_NOCONTRACT
  dets += blockIdx.z * bs.dets; order += blockIdx.z * bs.order; vote += blockIdx.z * bs.words; snap += blockIdx.z * bs.words;
  keep_q += blockIdx.z * bs.ints; out += blockIdx.z * bs.out;
  const int mrow = blockIdx.x;
  if (mrow >= d_nkeep[blockIdx.z]) return;
  HIP_DYNAMIC_SHARED(unsigned char, smem);  // (cap + 2) ints + (cap + 2) floats
  int* nbl = (int*)smem;
  float* yws = (float*)(smem + (size_t)(cap + 2) * 4);
  const int lane = threadIdx.x;
  const int K = d_count ? min(d_count[blockIdx.z], cap) : cap;
  const int nw = (K + 63) >> 6;
  const int q = keep_q[mrow];
  const int irow = order[q];
  const float yaw_i = dets[(size_t)irow * 12 + 8];
  // neighbourhood list in ascending sorted position, self first (nms.h:496-517)
  int n = 1;
  if (lane == 0) {
    nbl[0] = q;
    yws[0] = yaw_i;
  }
  for (int w = q >> 6; w < nw; ++w) {
    unsigned long long word = vote[(size_t)q * nwcap + w] & ~snap[(size_t)mrow * nwcap + w];
    if ((word >> lane) & 1ull) {
      int pos = n + __popcll(word & ((1ull << lane) - 1ull));
      int q2 = (w << 6) + lane;
      nbl[pos] = q2;
      yws[pos] = dets[(size_t)order[q2] * 12 + 8];
    }
    n += __popcll(word);
  }
  __syncthreads();
  // median yaw (nms.h:527-540)
  float med = yaw_i;
  if (n > 2) {
    int sz = n;
    if ((n & 1) == 0) {
      if (lane == 0) yws[n] = yaw_i;
      sz = n + 1;
    }
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
  if (lane < 11) out[(size_t)mrow * 12 + lane] = sum1 / sum3;
  if (lane == 11) out[(size_t)mrow * 12 + 11] = dets[(size_t)irow * 12 + 11];
}

__global__ __launch_bounds__(256) void iota_order_kernel(int* order, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) order[i] = i;
}

struct WnmsWs {
  float* prep;
  unsigned long long *thr, *vote, *snap, *supp_state;
  int *keep_q, *order, *alive, *nalive;
  void* sort_ws;
  int nwcap;
};
inline size_t wnms_ws_bytes(int cap);
inline WnmsWs wnms_ws_carve(void* ws, int cap);

}  // namespace rd
