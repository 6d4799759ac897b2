// bf16 implicit-GEMM convolution, second generation: weights never touch LDS.
//
// The first-generation kernel (k_conv.h, still used for fp32) stages BOTH MFMA operands in LDS with a 2x2 register tile
// per wave, which needs 1 KB of LDS reads per MFMA = 128 B/clk/CU at full MFMA rate -- exactly the LDS peak, so the
// matrix cores can never be more than ~half busy, and its weight ring costs a workgroup barrier per (chunk, tap) step.
// Here:
//   * every wave owns 128 output pixels (2 rows x 64 columns = four 32-px MFMA tiles) x 64 output channels (two 32-wide
//     tiles): 8 accumulators, 8 MFMAs per k-step of 16 channels;
//   * the pixel operand comes from the LDS halo tile as before (4 x ds_read_b128 per k-step = 0.5 KB / MFMA);
//   * the weight operand is read straight from global memory into registers.  The host packs the weights in MFMA
//     fragment order, so a fragment load is one fully coalesced 1 KB wave access (64 lanes x 16 B); the stream is
//     strictly sequential over the whole kernel, prefetched four k-steps ahead through a rotating register file
//     b[4][2], served by L1/L2 (every workgroup reads the same <= 295 KB);
//   * no barrier inside a k-chunk: workgroup barriers only bracket the halo restage (twice per 64-channel chunk).
// Same tap-list formulation, fused BN / ReLU / residual epilogue and transposed-MFMA output layout as k_conv.h.
#pragma once
#include "k_conv.h"

namespace rd {

// weight row held by A-operand lane row mm, so that the D registers of a lane are 16 consecutive output channels
__host__ __device__ inline int conv_row_perm(int mm) { return 16 * ((mm >> 2) & 1) + 4 * (mm >> 3) + (mm & 3); }

// k-steps of chunk c (16 channels each), and the padded count the kernel runs (multiple of the 4-deep register ring)
inline int wreg_nks(int nslots, int c) { return std::min(8, nslots - 8 * c) >> 1; }
inline int wreg_steps_padded(int ntaps, int nks) { return round_up(ntaps * nks, 4); }

// packed layout: [chunk][step u (tap-major, ks minor; zero steps pad each chunk to a multiple of 4)][cout/32][64 lanes][8 bf16]
template <class F>
inline void pack_taps_wreg(int ntaps, int cin, int cout, void* out, F get) {
  const int nslots = cin_slots(cin, RD_BF16), nchunk = (nslots + 7) / 8, ncb = cout / 32;
  bf16_t* o = (bf16_t*)out;
  for (int c = 0; c < nchunk; ++c) {
    const int nks = wreg_nks(nslots, c), nsc = ntaps * nks, nsp = wreg_steps_padded(ntaps, nks);
    for (int u = 0; u < nsp; ++u) {
      const int tap = u / nks, ks = u % nks;
      for (int cb = 0; cb < ncb; ++cb)
        for (int lane = 0; lane < 64; ++lane) {
          const int co = cb * 32 + conv_row_perm(lane & 31);
          for (int j = 0; j < 8; ++j) {
            const int ci = c * 64 + ks * 16 + (lane >> 5) * 8 + j;
            const float v = (u < nsc && ci < cin) ? get(co, ci, tap) : 0.f;
            *o++ = f32_to_bf16(v);
          }
        }
    }
  }
}

template <int WP, int WC>  // waves along the pixel rows (2 rows each) x waves along the output channels (64 each)
__global__ __launch_bounds__(WP * WC * 64, 2) void conv_wreg_kernel(ConvArgs a) {
  using T = bf16_t;
  HIP_DYNAMIC_SHARED(unsigned char, smem);
  constexpr int NW = WP * WC, NTH = NW * 64, RO = 2 * WP, COUT = 64 * WC;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave % WP, wcw = wave / WP;
  int widx;  // XCD-aware tile assignment (see k_conv.h)
  {
    const int Tn = gridDim.x, L = blockIdx.x, xcd = L & 7, i = L >> 3;
    const int qn = Tn >> 3, rn = Tn & 7;
    widx = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + i;
  }
  const int ct = widx % a.ncol;
  const int q0 = ct * 64, h0 = (widx / a.ncol) * RO, b = blockIdx.z;
  const int m = lane & 31, hi = lane >> 5;
  int tpt = 0;
#define RD_TRACE() { if (a.trace && tid == 0 && tpt < 8) a.trace[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8 + tpt++] = wall_clock64(); }
  RD_TRACE()

  unsigned char* As = smem;
  float* Sc = (float*)(smem + a.RI * a.CI * 128);  // [scale(COUT) | shift(COUT)]
  const T* x = (const T*)a.x + (size_t)b * a.x_bs;
  const int nchunk = (a.nslots + 7) >> 3;

  // halo tile: global -> registers -> LDS, all loads of a pass in flight together
  constexpr int AP = 4;
  auto a_stage = [&](int chunk) {
    const int ns_c = min(8, a.nslots - 8 * chunk);
    const int items = a.RI * a.CI * 8;
    for (int base = 0; base < items; base += NTH * AP) {
      Slot16 v[AP];
#pragma unroll
      for (int u = 0; u < AP; ++u) {
        const int idx = base + u * NTH + tid;
        const int px = idx >> 3, s = idx & 7;
        const int r = (int)__umulhi((unsigned)px, a.ci_magic), cc = px - r * a.CI;
        const int ih = h0 + a.min_dh + r, iw = q0 * a.in_stride + a.min_dw + cc;
        v[u] = Slot16{0u, 0u, 0u, 0u};
        if (idx < items && s < ns_c && ih >= 0 && ih < a.H && iw >= 0 && iw < a.Win && !(a.dbg & 16))
          v[u] = *(const Slot16*)(x + ((size_t)ih * a.Win + iw) * a.x_cs + a.x_co + (chunk * 8 + s) * 8);
      }
#pragma unroll
      for (int u = 0; u < AP; ++u) {
        const int idx = base + u * NTH + tid;
        const int px = idx >> 3, s = idx & 7;
        if (idx < items) *(Slot16*)(As + px * 128 + ((s ^ ((px >> 1) & 7)) << 4)) = v[u];
      }
    }
  };

  // weight fragment stream of this wave: step-major, two 1 KB fragments (its two 32-channel blocks) per step
  const int ncb = a.cout >> 5;
  const unsigned char* wbase = (const unsigned char*)a.w + (size_t)(wcw * 2) * 1024;   // wave-uniform
  const unsigned wlane = lane * 16;
  const int wstep = ncb * 1024;                                                       // bytes per k-step
  int nsteps_total = 0;
  for (int c = 0; c < nchunk; ++c) {
    const int nks = min(8, a.nslots - 8 * c) >> 1;
    nsteps_total += (a.ntaps * nks + 3) & ~3;
  }
  constexpr int DEPTH = 4;  // weight prefetch distance in k-steps (rotating register file)
  Slot16 bq[DEPTH][2];
  int wnext = 0;  // next step to fetch (clamped: the tail re-reads the last step, harmlessly)
#pragma unroll
  for (int k = 0; k < DEPTH; ++k) {
    const unsigned char* wp_ = wbase + (size_t)min(wnext, nsteps_total - 1) * wstep;
    bq[k][0] = *(const Slot16*)(wp_ + wlane);
    bq[k][1] = *(const Slot16*)(wp_ + wlane + 1024);
    ++wnext;
  }

  int pbase[4];  // halo pixel of this lane for tap (min_dh, min_dw), per 32-px tile: tile i = row 2*wp + (i>>1), column half i&1
#pragma unroll
  for (int i = 0; i < 4; ++i) pbase[i] = (2 * wp + (i >> 1)) * a.CI + ((i & 1) * 32 + m) * a.in_stride;
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (tid < COUT) {
    Sc[tid] = a.scale ? a.scale[tid] : 1.f;
    Sc[COUT + tid] = a.shift ? a.shift[tid] : 0.f;
  }

  // one k-step: 4 pixel fragments from LDS, 8 MFMAs against register-resident weights, refill the ring slot
#define RD_WREG_STEP(K, A0, KS)                                                                              \
  {                                                                                                          \
    const int kx_ = (KS) << 5;                                                                               \
    s16x8 av_[4];                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) av_[i] = *(const s16x8*)(smem + ((A0)[i] ^ kx_));         \
    s16x8 b0_, b1_;                                                                                          \
    memcpy(&b0_, &bq[(K) % DEPTH][0], 16);                                                                             \
    memcpy(&b1_, &bq[(K) % DEPTH][1], 16);                                                                             \
    if (!(a.dbg & 8)) _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                        \
      acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0_, av_[i], acc[i][0], 0, 0, 0);                  \
      acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1_, av_[i], acc[i][1], 0, 0, 0);                  \
    }                                                                                                        \
    if (!(a.dbg & 2)) {                                                                                      \
      const unsigned char* wp_ = wbase + (size_t)min(wnext, nsteps_total - 1) * wstep;                       \
      bq[(K) % DEPTH][0] = *(const Slot16*)(wp_ + wlane);                                                    \
      bq[(K) % DEPTH][1] = *(const Slot16*)(wp_ + wlane + 1024);                                             \
    }                                                                                                        \
    ++wnext;                                                                                                 \
    __builtin_amdgcn_sched_barrier(0); /* keep the refill HERE: hipcc otherwise sinks all loads to the loop end */ \
  }

  for (int chunk = 0; chunk < nchunk; ++chunk) {
    if (chunk == 0 || !(a.dbg & 4)) {
      if (chunk > 0) __syncthreads();  // every wave is done with the previous chunk's halo tile
      a_stage(chunk);
      __syncthreads();
    }
    RD_TRACE()
    const int nks = min(8, a.nslots - 8 * chunk) >> 1;
    if (nks == 4) {
      for (int tap = 0; tap < a.ntaps; ++tap) {
        const int tdh = (int)((a.dh_pack >> (4 * tap)) & 15) - 8, tdw = (int)((a.dw_pack >> (4 * tap)) & 15) - 8;
        const int delta = (tdh - a.min_dh) * a.CI + (tdw - a.min_dw);
        int a0[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int px = pbase[i] + delta;
          a0[i] = (px << 7) | ((((px >> 1) & 7) ^ hi) << 4);
        }
        RD_WREG_STEP(0, a0, 0)
        RD_WREG_STEP(1, a0, 1)
        RD_WREG_STEP(2, a0, 2)
        RD_WREG_STEP(3, a0, 3)
      }
    } else {
      // partial chunk (cin not a multiple of 64): steps are (tap, ks < nks) pairs, zero-weight steps pad to 4
      const int nsc = a.ntaps * nks, nsp = (nsc + 3) & ~3;
      for (int u = 0; u < nsp; u += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int uu = u + k;
          int tap = uu / nks, ks = uu - tap * nks;
          if (uu >= nsc) { tap = a.ntaps - 1; ks = 0; }
          const int tdh = (int)((a.dh_pack >> (4 * tap)) & 15) - 8, tdw = (int)((a.dw_pack >> (4 * tap)) & 15) - 8;
          const int delta = (tdh - a.min_dh) * a.CI + (tdw - a.min_dw);
          int a0[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int px = pbase[i] + delta;
            a0[i] = (px << 7) | ((((px >> 1) & 7) ^ hi) << 4);
          }
          if (k == 0) RD_WREG_STEP(0, a0, ks)
          if (k == 1) RD_WREG_STEP(1, a0, ks)
          if (k == 2) RD_WREG_STEP(2, a0, ks)
          if (k == 3) RD_WREG_STEP(3, a0, ks)
        }
      }
    }
    if (chunk + 1 < nchunk) RD_TRACE()
  }
#undef RD_WREG_STEP
  RD_TRACE()

  // ---- epilogue (layout as in k_conv.h): lane (px, hi) holds channels 64*wcw + 32*j + 16*hi + r of its pixel
  const bool relu_pre = a.flags & RD_RELU_PRE, do_add = a.flags & RD_ADD, relu_post = a.flags & RD_RELU_POST;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int oh = h0 + 2 * wp + (i >> 1);
    const int q = q0 + (i & 1) * 32 + m;
    const bool live = q < a.Wq && oh < a.H;
    const size_t pix = (size_t)oh * a.Wout + (size_t)q * a.out_stride + a.out_off;
    T* __restrict__ y = (T*)a.y + (size_t)b * a.y_bs + pix * a.y_cs + a.y_co + wcw * 64;
    const T* __restrict__ res = (const T*)a.res + (size_t)b * a.r_bs + pix * a.r_cs + a.r_co + wcw * 64;
    Slot16 rv[2][2];
    if (do_add && live) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int u = 0; u < 2; ++u) rv[j][u] = *(const Slot16*)(res + j * 32 + 16 * hi + u * 8);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int cb = j * 32 + 16 * hi;
      T rr[16];
      if (do_add && live) memcpy(rr, rv[j], sizeof(rr));
      T out[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 sc = *(const f32x4*)(Sc + wcw * 64 + cb + 4 * g);
        const f32x4 sh = *(const f32x4*)(Sc + COUT + wcw * 64 + cb + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          float v = acc[i][j][r] * sc[e] + sh[e];
          if (relu_pre) v = fmaxf(v, 0.f);
          if (do_add && live) v += Elem<RD_BF16>::to_f32(rr[r]);
          if (relu_post) v = fmaxf(v, 0.f);
          out[r] = Elem<RD_BF16>::from_f32(v);
        }
      }
      if (live && !(a.dbg & 1)) {
        Slot16 pk[2];
        memcpy(pk, out, sizeof(out));
        *(Slot16*)(y + cb) = pk[0];
        *(Slot16*)(y + cb + 8) = pk[1];
      }
    }
  }
  RD_TRACE()
#undef RD_TRACE
}

// dev tracing: RD_CONV_TRACE=1 allocates a device buffer the kernel stamps with s_memrealtime (100 MHz) per phase
constexpr size_t CONV_TRACE_CAP = 1 << 20;
inline unsigned long long* conv_trace_buf() {
#ifdef HIPEMU
  return nullptr;
#else
  static unsigned long long* buf = [] {
    unsigned long long* p = nullptr;
    if (getenv("RD_CONV_TRACE") && hipMalloc((void**)&p, CONV_TRACE_CAP * 8) == hipSuccess) (void)hipMemset(p, 0, CONV_TRACE_CAP * 8);
    return p;
  }();
  return buf;
#endif
}

// launch for the ConvArgs prepared by launch_conv (bf16 only)
inline int launch_conv_wreg(ConvArgs& a, int B, int H, int Wq, int mxdh, int mxdw, hipStream_t st) {
  constexpr int RO = 4;
  a.RI = RO + (mxdh - a.min_dh);
  a.CI = 63 * a.in_stride + (mxdw - a.min_dw) + 1;
  a.ncol = (Wq + 63) / 64;
  a.ci_magic = (unsigned)((1ull << 32) / (unsigned)a.CI) + 1u;
  const size_t lds = (size_t)a.RI * a.CI * 128 + 2 * a.cout * sizeof(float);
  RD_REQUIRE(lds <= 160 * 1024, RD_ESHAPE, "conv: LDS tile %zu B too large", lds);
  dim3 grid(a.ncol * ((H + RO - 1) / RO), 1, B);
  a.trace = nullptr;
  if (conv_trace_buf() && (size_t)grid.x * B * 8 <= CONV_TRACE_CAP) a.trace = conv_trace_buf();
  ProfScope ps(RD_PROF_CONV, st);
  if (a.cout == 128) hipLaunchKernelGGL((conv_wreg_kernel<2, 2>), grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((conv_wreg_kernel<2, 1>), grid, dim3(128), lds, st, a);
  return check_launch("conv_wreg_kernel");
}

}  // namespace rd
