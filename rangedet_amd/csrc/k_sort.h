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

__global__ __launch_bounds__(256) void sort_hist_kernel(const unsigned* __restrict__ keys, long N, int shift,
                                                        unsigned* __restrict__ hist, unsigned* __restrict__ tot,
                                                        long ws_stride) {
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
                                                           long ws_stride) {
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

// workspace: keysA, keysB, idxA, idxB (N each), hist (nblk*256), tot (4*256)
struct SortWs {
  unsigned *keysA, *keysB, *idxA, *idxB, *hist, *tot;
  int nblk;
};
inline size_t sort_ws_bytes(long N) {
  long nblk = (N + SORT_TILE - 1) / SORT_TILE;
  return (size_t)(4 * N + nblk * 256 + 4 * 256) * 4 + 256;
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
  s.tot = p;
  return s;
}
// sorts (keysA, idxA) ascending by key, stable; result back in keysA/idxA (4 passes)
// nb independent problems laid out ws_stride (in 4-byte words) apart in the workspace run as blockIdx.y
inline int radix_sort_pairs(const SortWs& s, long N, hipStream_t st, int nb = 1, long ws_stride = 0) {
  for (int b = 0; b < nb; ++b)
    if (hipMemsetAsync(s.tot + b * ws_stride, 0, 4 * 256 * 4, st) != hipSuccess) return fail(RD_EHIP, "sort: memset");
  unsigned *ki = s.keysA, *ko = s.keysB, *ii = s.idxA, *io = s.idxB;
  for (int pass = 0; pass < 4; ++pass) {
    hipLaunchKernelGGL(sort_hist_kernel, dim3(s.nblk, nb), dim3(256), 0, st, ki, N, pass * 8, s.hist, s.tot + pass * 256,
                       ws_stride);
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(s.nblk, nb), dim3(256), 0, st, ki, ii, N, pass * 8, s.hist,
                       s.tot + pass * 256, ko, io, ws_stride);
    std::swap(ki, ko);
    std::swap(ii, io);
  }
  return check_launch("radix_sort_pairs");
}

}  // namespace rd
