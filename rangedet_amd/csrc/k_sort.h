// get_sorted_foreground (operator_py/get_sorted_foreground.py:11-40) as a stable LSD radix sort on the GPU:
//   key = ~orderable_bits(score * mask)  -> ascending key order == descending score, and stability gives the
//   tie rule (score desc, flat index asc).  4 passes of 8 bits; per pass one histogram kernel and one
//   scatter kernel (the per-block digit prefix is summed inside the scatter kernel, digit totals by atomics),
//   then a gather of deltas / points.  All HBM-bound streaming over N*(4+4) bytes per pass.
#pragma once
#include "rd_common.h"

namespace rd {

constexpr int SORT_TILE = 2048;  // keys per workgroup: 4 waves x 8 rounds x 64 lanes

__device__ __forceinline__ unsigned desc_key(float f) {
  unsigned u = __float_as_uint(f);
  unsigned asc = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
  return ~asc;
}
__device__ __forceinline__ float key_to_float(unsigned key) {
  unsigned asc = ~key;
  unsigned u = (asc >> 31) ? (asc ^ 0x80000000u) : ~asc;
  return __uint_as_float(u);
}

__global__ __launch_bounds__(256) void sort_keygen_kernel(const float* __restrict__ score, const float* __restrict__ mask,
                                                          long N, int apply_sigmoid, unsigned* __restrict__ keys,
                                                          unsigned* __restrict__ idx, long ws_stride) {
  long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= N) return;
  const long b = blockIdx.y;                       // batch element: independent sort problems run side by side
  float s = score[b * N + i];
  if (apply_sigmoid) s = 1.0f / (1.0f + expf(-s));
  if (mask) s = s * mask[b * N + i];
  keys[b * ws_stride + i] = desc_key(s);
  idx[b * ws_stride + i] = (unsigned)i;
}
// keys for the weighted-NMS ordering: rows >= *d_count sort last
__global__ __launch_bounds__(256) void sort_keygen_dets_kernel(const float* __restrict__ dets, int cap,
                                                               const int* __restrict__ d_count, unsigned* __restrict__ keys,
                                                               unsigned* __restrict__ idx) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cap) return;
  int n = d_count ? min(*d_count, cap) : cap;
  keys[i] = i < n ? desc_key(dets[(size_t)i * 12 + 11]) : 0xFFFFFFFFu;
  idx[i] = (unsigned)i;
}

// d_n (optional): device-side element counts, one per problem and ws_stride words apart (the top-k pre-selection leaves M <= N
// candidates); blocks past the count only write their zero histogram
__global__ __launch_bounds__(256) void sort_hist_kernel(const unsigned* __restrict__ keys, long N, int shift,
                                                        unsigned* __restrict__ hist, unsigned* __restrict__ tot,
                                                        long ws_stride, const unsigned* __restrict__ d_n) {
  if (d_n) N = min(N, (long)d_n[blockIdx.y * ws_stride]);
  keys += blockIdx.y * ws_stride;
  hist += blockIdx.y * ws_stride;
  tot += blockIdx.y * ws_stride;
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  long base = (long)blockIdx.x * SORT_TILE;
  for (int r = 0; r < SORT_TILE / 256; ++r) {
    long i = base + r * 256 + threadIdx.x;
    if (i < N) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  unsigned v = h[threadIdx.x];
  hist[(size_t)blockIdx.x * 256 + threadIdx.x] = v;
  if (v) atomicAdd(&tot[threadIdx.x], v);
}

__global__ __launch_bounds__(256) void sort_scatter_kernel(const unsigned* __restrict__ keys_in,
                                                           const unsigned* __restrict__ idx_in, long N, int shift,
                                                           const unsigned* __restrict__ hist,
                                                           const unsigned* __restrict__ tot,
                                                           unsigned* __restrict__ keys_out, unsigned* __restrict__ idx_out,
                                                           long ws_stride, const unsigned* __restrict__ d_n) {
  if (d_n) {
    N = min(N, (long)d_n[blockIdx.y * ws_stride]);
    if ((long)blockIdx.x * SORT_TILE >= N) return;                 // (uniform: nothing to scatter)
  }
  {
    const long o = blockIdx.y * ws_stride;
    keys_in += o; idx_in += o; hist += o; tot += o; keys_out += o; idx_out += o;
  }
  __shared__ unsigned wcnt[4][256];
  __shared__ unsigned gbase[256];
  __shared__ unsigned scan[256];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0;
  // digit `tid`: exclusive prefix over digits (global totals) + keys of the same digit in earlier tiles
  {
    unsigned t = tot[tid];
    scan[tid] = t;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      unsigned a = tid >= off ? scan[tid - off] : 0u;
      __syncthreads();
      scan[tid] += a;
      __syncthreads();
    }
    unsigned before = 0;
    for (unsigned b = 0; b < blockIdx.x; ++b) before += hist[(size_t)b * 256 + tid];
    gbase[tid] = scan[tid] - t + before;
  }
  __syncthreads();

  unsigned key[8], id[8], pos[8];
  const long base = (long)blockIdx.x * SORT_TILE + wv * 512;
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    long i = base + r * 64 + lane;
    bool valid = i < N;
    key[r] = valid ? keys_in[i] : 0xFFFFFFFFu;
    id[r] = valid ? idx_in[i] : 0u;
    unsigned d = (key[r] >> shift) & 255u;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      unsigned long long bm = __ballot(valid && ((d >> bit) & 1u));
      peers &= ((d >> bit) & 1u) ? bm : ~bm;
    }
    unsigned rank = __popcll(peers & lt), cnt = __popcll(peers);
    unsigned prev = wcnt[wv][d];
    pos[r] = prev + rank;
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == 0) wcnt[wv][d] = prev + cnt;
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {  // per-digit exclusive prefix over the 4 waves
    unsigned c0 = wcnt[0][tid], c1 = wcnt[1][tid], c2 = wcnt[2][tid];
    wcnt[0][tid] = 0;
    wcnt[1][tid] = c0;
    wcnt[2][tid] = c0 + c1;
    wcnt[3][tid] = c0 + c1 + c2;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    long i = base + r * 64 + lane;
    if (i < N) {
      unsigned d = (key[r] >> shift) & 255u;
      unsigned dst = gbase[d] + wcnt[wv][d] + pos[r];
      keys_out[dst] = key[r];
      idx_out[dst] = id[r];
    }
  }
}

__global__ __launch_bounds__(256) void sort_gather_kernel(const unsigned* __restrict__ keys, const unsigned* __restrict__ idx,
                                                          long k, int D, const float* __restrict__ delta,
                                                          const float* __restrict__ pc, float* __restrict__ out_score,
                                                          float* __restrict__ out_delta, float* __restrict__ out_pc,
                                                          int* __restrict__ out_idx, long ws_stride, long N) {
  long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= k) return;
  {
    const long b = blockIdx.y;
    keys += b * ws_stride; idx += b * ws_stride;
    delta += b * N * D; pc += b * N * 3;
    out_score += b * k; out_delta += b * k * D; out_pc += b * k * 3;
    if (out_idx) out_idx += b * k;
  }
  unsigned j = idx[i];
  out_score[i] = key_to_float(keys[i]);
  for (int c = 0; c < D; ++c) out_delta[i * D + c] = delta[(size_t)j * D + c];
  out_pc[i * 3 + 0] = pc[(size_t)j * 3 + 0];
  out_pc[i * 3 + 1] = pc[(size_t)j * 3 + 1];
  out_pc[i * 3 + 2] = pc[(size_t)j * 3 + 2];
  if (out_idx) out_idx[i] = (int)j;
}

// ---- top-k pre-selection ---------------------------------------------------------------------------------------------
// get_sorted_foreground keeps the k best of N scores (50 000 of 297 472): instead of sorting all N keys, one histogram over
// the keys' top 12 bits finds the threshold bin T (the smallest T with at least k keys in bins <= T), the keys of bins <= T
// are compacted in index order (M >= k of them, typically k + one bin's population) and only those are sorted.  The first k
// of the result are exactly the first k of the full sort: every key of a bin < T is smaller than every key of bin T, ties
// keep their index order through the ordered compaction and the stable LSD passes.
constexpr int SEL_BINS = 4096, SEL_SHIFT = 20;
__global__ __launch_bounds__(256) void select_hist_kernel(const unsigned* __restrict__ keys, long N, unsigned* __restrict__ h12,
                                                          long ws_stride) {
  keys += blockIdx.y * ws_stride; h12 += blockIdx.y * ws_stride;
  __shared__ unsigned h[SEL_BINS];
  for (int i = threadIdx.x; i < SEL_BINS; i += 256) h[i] = 0;
  __syncthreads();
  const long base = (long)blockIdx.x * SORT_TILE;
  for (int r = 0; r < SORT_TILE / 256; ++r) {
    const long i = base + r * 256 + threadIdx.x;
    if (i < N) atomicAdd(&h[keys[i] >> SEL_SHIFT], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SEL_BINS; i += 256)
    if (h[i]) atomicAdd(&h12[i], h[i]);
}
// one workgroup per problem: sel[0] = threshold bin T, sel[1] = M = number of keys in bins <= T
__global__ __launch_bounds__(256) void select_thresh_kernel(const unsigned* __restrict__ h12, long k, unsigned* __restrict__ sel,
                                                            long ws_stride) {
  h12 += blockIdx.y * ws_stride; sel += blockIdx.y * ws_stride;
  __shared__ unsigned part[256];
  const int tid = threadIdx.x;
  unsigned loc[SEL_BINS / 256], sum = 0;
#pragma unroll
  for (int j = 0; j < SEL_BINS / 256; ++j) { loc[j] = h12[tid * (SEL_BINS / 256) + j]; sum += loc[j]; }
  part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const unsigned v = tid >= off ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  unsigned before = part[tid] - sum;                               // keys in the bins of earlier threads
  // fallback when no bin reaches k (k > number of keys, or k == 0 -- both rejected by the callers): every key is a candidate, so
  // the later passes never read an uninitialised threshold
  if (tid == 255 && (k == 0 || part[255] < (unsigned)k)) { sel[0] = SEL_BINS - 1; sel[1] = part[255]; }
  if (before < (unsigned)k && before + sum >= (unsigned)k) {       // exactly one thread: the threshold bin is one of its 16
#pragma unroll
    for (int j = 0; j < SEL_BINS / 256; ++j) {
      if (before < (unsigned)k && before + loc[j] >= (unsigned)k) { sel[0] = tid * (SEL_BINS / 256) + j; sel[1] = before + loc[j]; }
      before += loc[j];
    }
  }
}
// candidates per SORT_TILE block
__global__ __launch_bounds__(256) void select_count_kernel(const unsigned* __restrict__ keys, long N, const unsigned* __restrict__ sel,
                                                           unsigned* __restrict__ cnt, long ws_stride) {
  keys += blockIdx.y * ws_stride; sel += blockIdx.y * ws_stride; cnt += blockIdx.y * ws_stride;
  const unsigned T = sel[0];
  __shared__ unsigned c;
  if (threadIdx.x == 0) c = 0;
  __syncthreads();
  const long base = (long)blockIdx.x * SORT_TILE;
  unsigned mine = 0;
  for (int r = 0; r < SORT_TILE / 256; ++r) {
    const long i = base + r * 256 + threadIdx.x;
    if (i < N && (keys[i] >> SEL_SHIFT) <= T) ++mine;
  }
  if (mine) atomicAdd(&c, mine);
  __syncthreads();
  if (threadIdx.x == 0) cnt[blockIdx.x] = c;
}
// ordered compaction: position = candidates in earlier blocks + earlier waves' rounds + rank inside the wave's round
__global__ __launch_bounds__(256) void select_compact_kernel(const unsigned* __restrict__ keys_in, const unsigned* __restrict__ idx_in,
                                                             long N, const unsigned* __restrict__ sel,
                                                             const unsigned* __restrict__ cnt, unsigned* __restrict__ keys_out,
                                                             unsigned* __restrict__ idx_out, long ws_stride) {
  {
    const long o = blockIdx.y * ws_stride;
    keys_in += o; idx_in += o; sel += o; cnt += o; keys_out += o; idx_out += o;
  }
  const unsigned T = sel[0];
  __shared__ unsigned wcnt[4][8], sbase;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) {
    unsigned b = 0;
    for (unsigned j = 0; j < blockIdx.x; ++j) b += cnt[j];
    sbase = b;
  }
  // element order inside a block: wave wv owns 512 consecutive keys, in 8 rounds of 64 (like sort_scatter_kernel)
  const long base = (long)blockIdx.x * SORT_TILE + wv * 512;
  unsigned key[8], id[8];
  unsigned long long m[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const long i = base + r * 64 + lane;
    const bool valid = i < N;
    key[r] = valid ? keys_in[i] : 0xFFFFFFFFu;
    id[r] = valid ? idx_in[i] : 0u;
    m[r] = __ballot(valid && (key[r] >> SEL_SHIFT) <= T);
    if (lane == 0) wcnt[wv][r] = (unsigned)__popcll(m[r]);
  }
  __syncthreads();
  unsigned off = sbase;
  for (int w = 0; w < wv; ++w)
#pragma unroll
    for (int r = 0; r < 8; ++r) off += wcnt[w][r];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    if ((m[r] >> lane) & 1ull) {
      const unsigned dst = off + (unsigned)__popcll(m[r] & ((1ull << lane) - 1ull));
      keys_out[dst] = key[r];
      idx_out[dst] = id[r];
    }
    off += (unsigned)__popcll(m[r]);
  }
}

// workspace: keysA, keysB, idxA, idxB (N each), hist (nblk*256), tot (4*256), then the pre-selection's h12 (4096), block counts
// (nblk) and (T, M)
struct SortWs {
  unsigned *keysA, *keysB, *idxA, *idxB, *hist, *tot, *h12, *cnt, *sel;
  int nblk;
};
inline size_t sort_ws_bytes(long N) {
  long nblk = (N + SORT_TILE - 1) / SORT_TILE;
  return (size_t)(4 * N + nblk * 256 + 4 * 256 + SEL_BINS + nblk + 64) * 4 + 256;
}
inline SortWs sort_ws_carve(void* ws, long N) {
  SortWs s;
  s.nblk = (int)((N + SORT_TILE - 1) / SORT_TILE);
  unsigned* p = (unsigned*)ws;
  s.keysA = p; p += N;
  s.keysB = p; p += N;
  s.idxA = p; p += N;
  s.idxB = p; p += N;
  s.hist = p; p += (size_t)s.nblk * 256;
  s.tot = p; p += 4 * 256;
  s.h12 = p; p += SEL_BINS;
  s.cnt = p; p += s.nblk;
  s.sel = p;
  return s;
}
// sorts (keysA, idxA) ascending by key, stable; result back in keysA/idxA (4 passes)
// nb independent problems laid out ws_stride (in 4-byte words) apart in the workspace run as blockIdx.y
// zero the digit totals (and, with_h12, the pre-selection histogram behind them) of all nb problems with ONE fill
inline int sort_clear(const SortWs& s, hipStream_t st, int nb, long ws_stride, bool with_h12) {
  const size_t bytes = (size_t)(4 * 256 + (with_h12 ? SEL_BINS : 0)) * 4;
  const hipError_t e = nb > 1 ? hipMemset2DAsync(s.tot, (size_t)ws_stride * 4, 0, bytes, nb, st) : hipMemsetAsync(s.tot, 0, bytes, st);
  return e == hipSuccess ? RD_OK : fail(RD_EHIP, "sort: memset");
}
// d_n (optional): device-side element counts (see sort_hist_kernel)
inline int radix_sort_pairs(const SortWs& s, long N, hipStream_t st, int nb = 1, long ws_stride = 0, const unsigned* d_n = nullptr,
                            bool cleared = false) {
  if (!cleared) { int rc = sort_clear(s, st, nb, ws_stride, false); if (rc != RD_OK) return rc; }
  unsigned *ki = s.keysA, *ko = s.keysB, *ii = s.idxA, *io = s.idxB;
  for (int pass = 0; pass < 4; ++pass) {
    hipLaunchKernelGGL(sort_hist_kernel, dim3(s.nblk, nb), dim3(256), 0, st, ki, N, pass * 8, s.hist, s.tot + pass * 256,
                       ws_stride, d_n);
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(s.nblk, nb), dim3(256), 0, st, ki, ii, N, pass * 8, s.hist,
                       s.tot + pass * 256, ko, io, ws_stride, d_n);
    std::swap(ki, ko);
    std::swap(ii, io);
  }
  return check_launch("radix_sort_pairs");
}

// the k smallest (key, index) pairs of keysA / idxA, sorted: in *rk / *ri (one of the two buffer pairs; positions >= k unspecified)
inline int radix_topk_pairs(const SortWs& s, long N, long k, hipStream_t st, int nb, long ws_stride, unsigned** rk, unsigned** ri) {
  const bool off = dev_switches().sort_no_select;      // dev switch
  *rk = s.keysA; *ri = s.idxA;
  RD_REQUIRE(k > 0 && k <= N, RD_ESHAPE, "top-k: k %ld of N %ld", k, N);
  if (off || 2 * k > N) return radix_sort_pairs(s, N, st, nb, ws_stride);
  { int rc = sort_clear(s, st, nb, ws_stride, true); if (rc != RD_OK) return rc; }
  hipLaunchKernelGGL(select_hist_kernel, dim3(s.nblk, nb), dim3(256), 0, st, s.keysA, N, s.h12, ws_stride);
  hipLaunchKernelGGL(select_thresh_kernel, dim3(1, nb), dim3(256), 0, st, s.h12, k, s.sel, ws_stride);
  hipLaunchKernelGGL(select_count_kernel, dim3(s.nblk, nb), dim3(256), 0, st, s.keysA, N, s.sel, s.cnt, ws_stride);
  hipLaunchKernelGGL(select_compact_kernel, dim3(s.nblk, nb), dim3(256), 0, st, s.keysA, s.idxA, N, s.sel, s.cnt, s.keysB, s.idxB,
                     ws_stride);
  // sort the M candidates (in keysB / idxB): four passes B -> A -> B -> A -> B, so start from a swapped view
  SortWs v = s;
  v.keysA = s.keysB; v.keysB = s.keysA; v.idxA = s.idxB; v.idxB = s.idxA;
  *rk = s.keysB; *ri = s.idxB;                                     // (an even number of passes ends where it started)
  return radix_sort_pairs(v, N, st, nb, ws_stride, s.sel + 1, true);
}

}  // namespace rd
