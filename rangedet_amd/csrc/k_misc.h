// Bandwidth-bound helpers of the hot path: layout conversion at the NCHW float32 boundary, the 1x1 head output
// convs (logits / deltas, flattened), Decode3DBbox, the score filter + 10->11-dim conversion and 12->8-dim.
#pragma once
#include "rd_common.h"

namespace rd {

// ---- NCHW f32 <-> channels-last -------------------------------------------------------------------------
// One thread per (pixel, channel-slot): reads are strided by H*W per channel (each channel plane is read
// coalesced along W by consecutive threads), writes are contiguous 16-byte slots.
template <int DT>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, void* __restrict__ dst_, int C,
                                                           long HW, int cs, int coff, int cpad, long total) {
  using E = Elem<DT>;
  typename E::T* dst = (typename E::T*)dst_;
  const int ctot = C + cpad;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long p = i % HW;          // pixel within image (fastest across threads -> coalesced plane reads)
    long rest = i / HW;
    int c = (int)(rest % ctot);
    long b = rest / ctot;
    float v = c < C ? src[(b * C + c) * HW + p] : 0.f;
    dst[(b * HW + p) * cs + coff + c] = E::from_f32(v);
  }
}
// few channels (the 8-channel range image): one thread per pixel gathers its channels (plane reads stay coalesced across
// the threads) and writes them as whole 16-byte slots -- the per-element kernel above writes 2 bytes per thread.
template <int DT, int NSLOT>   // NSLOT 16-byte slots = C + zero padding
__global__ __launch_bounds__(256) void nchw_to_nhwc_px_kernel(const float* __restrict__ src, void* __restrict__ dst_, int C,
                                                              long HW, int cs, int coff, long npix) {
  using E = Elem<DT>;
  typename E::T* dst = (typename E::T*)dst_;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    const long b = i / HW, p = i - b * HW;
    typename E::T v[NSLOT * E::CH];
#pragma unroll
    for (int c = 0; c < NSLOT * E::CH; ++c) v[c] = E::from_f32(c < C ? src[(b * C + c) * HW + p] : 0.f);
    Slot16 pk[NSLOT];
    memcpy(pk, v, sizeof(pk));
#pragma unroll
    for (int u = 0; u < NSLOT; ++u) *(Slot16*)(dst + i * cs + coff + u * E::CH) = pk[u];
  }
}
template <int DT>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const void* __restrict__ src_, float* __restrict__ dst, int C,
                                                           long HW, int cs, int coff, long total) {
  using E = Elem<DT>;
  const typename E::T* src = (const typename E::T*)src_;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long p = i % HW;
    long rest = i / HW;
    int c = (int)(rest % C);
    long b = rest / C;
    dst[(b * C + c) * HW + p] = E::to_f32(src[(b * HW + p) * cs + coff + c]);
  }
}

// ---- head output 1x1 convs ---------------------------------------------------------------------------------
// 8 lanes cooperate on one pixel (each lane owns cin/8 channels, contiguous), so a wave reads 8 whole pixels
// = 8 * cin * sizeof(T) contiguous bytes per iteration; partial dot products are reduced with 3 xor-shuffles.
template <int DT, int NOUT>
__global__ __launch_bounds__(256) void head_out_kernel(const void* __restrict__ x_, int cs, int coff,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ out, long out_bs, long n_off, long HW,
                                                       int cin) {
  using E = Elem<DT>;
  using T = typename E::T;
  __shared__ float wl[8 * 128];
  const T* x = (const T*)x_ + (size_t)blockIdx.y * HW * cs;
  for (int i = threadIdx.x; i < NOUT * cin; i += 256) wl[i] = w[i];
  __syncthreads();
  const int sub = threadIdx.x & 7;          // which 1/8 of the channels
  const int per = cin >> 3;                 // channels per lane (cin % 8 == 0)
  const long npix_iter = (long)gridDim.x * 32;
  for (long p0 = blockIdx.x * 32L; p0 < HW; p0 += npix_iter) {
    long p = p0 + (threadIdx.x >> 3);
    float acc[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) acc[o] = 0.f;
    if (p < HW) {
      const T* px = x + p * cs + coff + sub * per;
      if ((per % E::CH) == 0) {             // 16-byte vector loads (the normal case: per = 16 channels)
        for (int c0 = 0; c0 < per; c0 += E::CH) {
          const Slot16 raw = *(const Slot16*)(px + c0);
          T el[E::CH];
          memcpy(el, &raw, 16);
#pragma unroll
          for (int e = 0; e < E::CH; ++e) {
            const float v = E::to_f32(el[e]);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) acc[o] += v * wl[o * cin + sub * per + c0 + e];
          }
        }
      } else {
        for (int c = 0; c < per; ++c) {
          float v = E::to_f32(px[c]);
#pragma unroll
          for (int o = 0; o < NOUT; ++o) acc[o] += v * wl[o * cin + sub * per + c];
        }
      }
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      acc[o] += __shfl_xor(acc[o], 1);
      acc[o] += __shfl_xor(acc[o], 2);
      acc[o] += __shfl_xor(acc[o], 4);
    }
    if (p < HW && sub == 0) {
      float* o_ = out + (size_t)blockIdx.y * out_bs + (n_off + p) * NOUT;
#pragma unroll
      for (int o = 0; o < NOUT; ++o) o_[o] = acc[o] + bias[o];
    }
  }
}

// bf16 variant on the matrix cores: the op is a [NOUT x cin] x [cin x pixels] GEMM streamed once over the feature map
// (HBM-bound: cin*2 bytes in, NOUT*4 bytes out per pixel).  One wave owns 32 pixels at a time.  Global loads are whole
// pixel rows (a wave instruction = 4 pixels x 256 B contiguous), transposed through a wave-private, XOR-swizzled 8 KB
// LDS image into the MFMA B-operand layout (lane (px, hi) <- 16-byte channel group 2*ks + hi of its pixel).  The A
// operand is the weight matrix padded to 32 rows, held in registers as a bf16 hi + lo pair per k-step (w = hi + lo to
// ~2^-17, so fp32 weights keep their precision); D[co][px] leaves lane (px, hi) with co = 4*hi + r in registers r = 0..3.
template <int NOUT, int DT = RD_BF16>
__global__ __launch_bounds__(256) void head_out_mfma_kernel(const bf16_t* __restrict__ x, int cs, int coff,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ out, long out_bs, long n_off, long HW,
                                                            int cin) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 8192];
  const int lane = threadIdx.x & 63, m = lane & 31, hi = lane >> 5, wv = threadIdx.x >> 6;
  unsigned char* img = lds + wv * 8192;
  const int nks = cin >> 4, nsl = cin >> 3;                     // k-steps (<= 8), 16-byte slots per pixel (<= 16)
  x += (size_t)blockIdx.y * HW * cs + coff;
  out += (size_t)blockIdx.y * out_bs + n_off * NOUT;
  s16x8 wh[8], wl[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    unsigned short h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = (m < NOUT && ks < nks) ? w[m * cin + ks * 16 + hi * 8 + j] : 0.f;
      h[j] = H16<DT>::from_f32(v);
      l[j] = H16<DT>::from_f32(v - H16<DT>::to_f32(h[j]));
    }
    memcpy(&wh[ks], h, 16);
    memcpy(&wl[ks], l, 16);
  }
  float bs[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bs[r] = 4 * hi + r < NOUT ? bias[4 * hi + r] : 0.f;
  const long ntile = (HW + 31) / 32;
  const long wave0 = (long)blockIdx.x * 4 + wv, nwave = (long)gridDim.x * 4;
  const int lp = lane >> 4, ls = lane & 15;                     // load role: pixel it*4 + lp of the tile, slot ls
  auto load = [&](long tile, Slot16 (&v)[8]) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const long p = min(tile * 32 + it * 4 + lp, HW - 1);      // clamp: dead pixels re-read the last one, never stored
      v[it] = ls < nsl ? *(const Slot16*)(x + p * cs + ls * 8) : Slot16{0u, 0u, 0u, 0u};
    }
  };
  Slot16 cur[8], nxt[8];
  if (wave0 < ntile) load(wave0, cur);
  for (long tile = wave0; tile < ntile; tile += nwave) {
    const bool more = tile + nwave < ntile;
    if (more) load(tile + nwave, nxt);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int pr = it * 4 + lp;
      *(Slot16*)(img + pr * 256 + ((ls ^ (pr & 15)) << 4)) = cur[it];
    }
    __builtin_amdgcn_wave_barrier();                            // LDS ops of a wave are in order; this orders hipemu's lanes
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const s16x8 b = *(const s16x8*)(img + m * 256 + (((2 * ks + hi) ^ (m & 15)) << 4));
      acc = H16<DT>::mfma(wl[ks], b, acc);
      acc = H16<DT>::mfma(wh[ks], b, acc);
    }
    __builtin_amdgcn_wave_barrier();
    const long p = tile * 32 + m;
    if (p < HW) {
      float* o = out + p * NOUT + 4 * hi;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * hi + r < NOUT) o[r] = acc[r] + bs[r];
    }
    if (more) {
#pragma unroll
      for (int it = 0; it < 8; ++it) cur[it] = nxt[it];
    }
  }
}

// ---- Decode3DBbox  (decode_3d_bbox-inl.h:64-277) ------------------------------------------------------------
// One thread per point: 44 B in, 40 B out.  Same operation order as the reference's Map(); the two spots where
// the reference's float instantiation goes through double (0.5*length, height/2.0) do so here too.
__global__ __launch_bounds__(256) void decode3d_kernel(const float* __restrict__ delta, const float* __restrict__ pc,
                                                       float* __restrict__ out, long n, int box_type, int is_bin) {
  long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= n) return;
  const float* d = delta + idx * box_type;
  float pc_x = pc[idx * 3 + 0], pc_y = pc[idx * 3 + 1], pc_z = pc[idx * 3 + 2];
  float az = atan2f(pc_y, pc_x);
  float ca = cosf(az), sa = sinf(az);
  float dx, dy, width, length, height, z0, yaw_l;
  if (is_bin) {
    dx = d[0];
    dy = d[1];
    width = expf(d[3]);
    length = expf(d[4]);
    height = expf(d[5]);
    float cz = pc_z + d[2];
    z0 = (float)((double)cz - (double)height / 2.0);
    yaw_l = d[6] + az;
  } else {
    dx = d[0] * fabsf(d[0]);
    dy = d[1] * fabsf(d[1]);
    width = expf(d[2]);
    length = expf(d[3]);
    height = expf(d[7]);
    z0 = d[6];
    yaw_l = atan2f(d[5], d[4]) + az;
  }
  float dxl = dx * ca - dy * sa;
  float dyl = dx * sa + dy * ca;
  float cx = pc_x + dxl, cy = pc_y + dyl;
  float sy = sinf(yaw_l), cyw = cosf(yaw_l);
  float hl = (float)(0.5 * (double)length), hw = (float)(0.5 * (double)width);
  float hx[4] = {hl, -hl, -hl, hl};
  float hy[4] = {-hw, -hw, hw, hw};
  float* o = out + idx * 10;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[2 * k] = (hx[k] * cyw - hy[k] * sy) + cx;
    o[2 * k + 1] = (hx[k] * sy + hy[k] * cyw) + cy;
  }
  o[8] = z0;
  o[9] = z0 + height;
}

// ---- score filter + 10->11 dim  (tools/test.py:56-81,200-209) --------------------------------------------------
// stable compaction of rows with score > min_score: per-block counts -> single-block scan -> scatter.
// batch: blockIdx.y selects the frame; per-frame buffers sit `*_bs` elements apart
__global__ __launch_bounds__(256) void filter_count_kernel(const float* __restrict__ scores, long n, float thr,
                                                           int* __restrict__ blk_cnt, long sc_bs, long blk_bs) {
  scores += blockIdx.y * sc_bs;
  blk_cnt += blockIdx.y * blk_bs;
  long i = blockIdx.x * 256L + threadIdx.x;
  int p = (i < n) && (scores[i] > thr);
  unsigned long long m = __ballot(p);
  __shared__ int wc[4];
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}
__global__ __launch_bounds__(256) void filter_scan_kernel(int* __restrict__ blk_cnt, int nblk, int* __restrict__ total,
                                                          long blk_bs) {
  // exclusive scan of nblk ints by one workgroup (nblk is a few hundred); one workgroup per frame
  blk_cnt += blockIdx.x * blk_bs;
  total += blockIdx.x;
  __shared__ int part[256];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += 256) {
    int i = base + threadIdx.x;
    int v = i < nblk ? blk_cnt[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      int t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    int incl = part[threadIdx.x];
    if (i < nblk) blk_cnt[i] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(256) void filter_scatter_kernel(const float* __restrict__ scores,
                                                             const float* __restrict__ boxes10, long n, float thr,
                                                             const int* __restrict__ blk_off, float* __restrict__ dets,
                                                             long sc_bs, long box_bs, long blk_bs, long dets_bs) {
  scores += blockIdx.y * sc_bs;
  boxes10 += blockIdx.y * box_bs;
  blk_off += blockIdx.y * blk_bs;
  dets += blockIdx.y * dets_bs;
  long i = blockIdx.x * 256L + threadIdx.x;
  int p = (i < n) && (scores[i] > thr);
  unsigned long long m = __ballot(p);
  __shared__ int wc[4];
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  if (ln == 0) wc[wv] = __popcll(m);
  __syncthreads();
  int off = blk_off[blockIdx.x];
  for (int w = 0; w < wv; ++w) off += wc[w];
  if (p) {
    int pos = off + __popcll(m & ((1ull << ln) - 1ull));
    const float* b = boxes10 + i * 10;
    float* d = dets + (size_t)pos * 12;
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = b[k];
    d[8] = fdlibm_atan2f(b[1] - b[3], b[0] - b[2]);  // yaw = arctan2(yA - yB, xA - xB): numpy's float32 arctan2 = the C library's atan2f where numpy has no SIMD routine (rd_common.h)
    d[9] = b[8];
    d[10] = b[9] - b[8];
    d[11] = scores[i];
  }
}

// ---- 12 -> 8 dim  (tools/test.py:43-53; float32 here, the reference's numpy promotes to float64) ---------------
__global__ __launch_bounds__(256) void dets12_to_8_kernel(const float* __restrict__ d12, int cap, const int* __restrict__ d_count,
                                                          float* __restrict__ o8, long d12_bs, long o8_bs) {
  d12 += blockIdx.y * d12_bs;
  o8 += blockIdx.y * o8_bs;
  int i = blockIdx.x * 256 + threadIdx.x;
  int n = d_count ? min(d_count[blockIdx.y], cap) : cap;
  if (i >= n) return;
  const float* d = d12 + (size_t)i * 12;
  float* o = o8 + (size_t)i * 8;
  o[0] = (d[0] + d[2] + d[4] + d[6]) / 4.f;
  o[1] = (d[1] + d[3] + d[5] + d[7]) / 4.f;
  o[2] = d[9] + d[10] / 2.f;
  float ax = d[2] - d[0], ay = d[3] - d[1];
  float bx = d[2] - d[4], by = d[3] - d[5];
  o[3] = sqrtf(ax * ax + ay * ay);
  o[4] = sqrtf(bx * bx + by * by);
  o[5] = d[10];
  o[6] = d[8];
  o[7] = d[11];
}

// ---- scores of the boxes NMS3D kept (tools/test.py:193-196: cls_score[keep_inds] on the valid entries); padding entries
// (keep = -1) get -inf so that the score filter that follows drops them ------------------------------------------------
__global__ __launch_bounds__(256) void gather_keep_scores_kernel(const float* __restrict__ score, long score_bs, int k,
                                                                 const int* __restrict__ keep, int mk, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= mk) return;
  const int q = keep[(size_t)blockIdx.y * mk + i];
  out[(size_t)blockIdx.y * mk + i] = (q >= 0 && q < k) ? score[blockIdx.y * score_bs + q] : -__builtin_inff();
}

}  // namespace rd
