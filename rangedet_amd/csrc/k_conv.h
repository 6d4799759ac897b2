// Implicit-GEMM convolution over channels-last range images for gfx950 (MFMA), with the BatchNorm affine,
// ReLU and residual add fused into the epilogue.  One kernel serves every conv-shaped layer of the RangeDet
// graph through a *tap list*:  y[h, q*out_stride+out_off, co] = sum_t sum_ci x[h+dh_t, q*in_stride+dw_t, ci] * W_t[ci][co]
//   3x3 / 1x1 convs, stride (1,1) or (1,2)        dla_backbone.py:18-56, head/builder.py:221-240 (reference)
//   transposed convs k(3,8)s(1,4) / k(3,4)s(1,2)  dla_backbone.py:117-127  -> one launch per output phase
//
// Work decomposition (wave64, 256-thread workgroups):
//   workgroup tile = RO output rows x 64 output columns x all Cout; each wave owns 64 px (one row, two 32-px
//   MFMA tiles) x 64 output channels (two 32-wide tiles) -> 4 accumulators of 32x32.
//   The input halo tile (RO+2 rows x 66 columns x one 128-byte k-chunk) is staged ONCE per k-chunk in LDS and
//   re-read by all taps (9x reuse for a 3x3); the per-(chunk,tap) weight slab [Cout][128 B] is double-buffered
//   in LDS (global loads issued before the MFMA block, LDS write after it).
//   LDS rows are 128 B (= 8 slots of 16 B = 64 bf16 / 32 f32 channels); slot index is XOR-swizzled with
//   ((row>>1)&7) so the 16-lane groups of ds_read_b128 hit 16 distinct slots (conflict-free for stride-1 pixels).
//   MFMA operands: lanes 0-31 read slot 2*ks, lanes 32-63 slot 2*ks+1 of their pixel / output channel:
//     bf16: that IS the v_mfma_f32_32x32x16_bf16 fragment (k = 8*(lane>>5)+j);
//     f32 : 4 x v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain), element e of the slot pairs k = 4*(2ks+hi)+e.
#pragma once
#include <cstdlib>

#include "rd_common.h"

namespace rd {

struct ConvArgs {
  const void* x; int x_cs, x_co; long x_bs;
  const void* w;
  const float* scale; const float* shift;
  const void* res; int r_cs, r_co; long r_bs;
  void* y; int y_cs, y_co; long y_bs;
  int H, Win, Wq, Wout;
  int nslots, cout, ntaps;
  int in_stride, out_stride, out_off;
  int min_dh, min_dw, RI, CI;
  int flags;
  unsigned long long dh_pack, dw_pack;  // 4 bits per tap, biased by 8 (no dynamically indexed kernarg arrays)
  int dbg;  // tuning ablations (tools/conv_bench.py): 1 no epilogue stores, 2 one weight slab only, 4 one halo stage, 8 no MFMA
};

template <int DT, int NT>  // NT = Cout / 32 output-channel tiles per wave (2 or 4)
__global__ __launch_bounds__(256, 2) void conv_taps_kernel(ConvArgs a) {
  using E = Elem<DT>;
  using T = typename E::T;
  HIP_DYNAMIC_SHARED(unsigned char, smem);
  constexpr int RO = 4;              // output rows per workgroup = one per wave
  constexpr int WCNT = NT;           // 16-byte slots of one weight slab handled per thread (Cout*8/256)
  const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
  const int q0 = blockIdx.x * 64, h0 = blockIdx.y * RO, b = blockIdx.z;
  const int m = lane & 31, hi = lane >> 5;

  unsigned char* As = smem;
  unsigned char* Ws = smem + a.RI * a.CI * 128;  // ONE weight slab [Cout][128 B]; the pipeline depth lives in registers
  const T* x = (const T*)a.x + (size_t)b * a.x_bs;
  const int nchunk = (a.nslots + 7) >> 3;
  const int nsteps = nchunk * a.ntaps;

  int b_row[NT], b_swz[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int n = nt * 32 + m;
    b_row[nt] = n * 128;
    b_swz[nt] = (n >> 1) & 7;
  }

  // weight slabs: global -> registers two steps ahead -> LDS right before use
  Slot16 wA[WCNT], wB[WCNT];
#define RD_W_LOAD(dst_, step_)                                                        \
  {                                                                                   \
    const Slot16* src_ = (const Slot16*)a.w + (size_t)(step_) * (NT * 32 * 8);        \
    _Pragma("unroll") for (int i_ = 0; i_ < WCNT; ++i_) dst_[i_] = src_[tid + i_ * 256]; \
  }
#define RD_W_STORE(src_)                                                              \
  {                                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < WCNT; ++i_) {                             \
      const int idx_ = tid + i_ * 256, c_ = idx_ >> 3, s_ = idx_ & 7;                 \
      const int n_ = (c_ & ~31) | ((c_ & 3) + 8 * ((c_ >> 2) & 3) + 4 * ((c_ >> 4) & 1)); \
      *(Slot16*)(Ws + n_ * 128 + ((s_ ^ ((n_ >> 1) & 7)) << 4)) = src_[i_];           \
    }                                                                                 \
  }
  // halo tile: all global loads of a pass are issued before the first LDS write (latency paid once per pass)
  constexpr int AP = 7;
  auto a_stage = [&](int chunk) {
    const int ns_c = min(8, a.nslots - 8 * chunk);
    const int items = a.RI * a.CI * 8;
    for (int base = 0; base < items; base += 256 * AP) {
      Slot16 v[AP];
#pragma unroll
      for (int u = 0; u < AP; ++u) {
        const int idx = base + u * 256 + tid;
        const int px = idx >> 3, s = idx & 7;
        const int r = px / a.CI, cc = px - r * a.CI;
        const int ih = h0 + a.min_dh + r, iw = q0 * a.in_stride + a.min_dw + cc;
        v[u] = Slot16{0u, 0u, 0u, 0u};
        if (idx < items && s < ns_c && ih >= 0 && ih < a.H && iw >= 0 && iw < a.Win)
          v[u] = *(const Slot16*)(x + ((size_t)ih * a.Win + iw) * a.x_cs + a.x_co + (chunk * 8 + s) * E::CH);
      }
#pragma unroll
      for (int u = 0; u < AP; ++u) {
        const int idx = base + u * 256 + tid;
        const int px = idx >> 3, s = idx & 7;
        if (idx < items) *(Slot16*)(As + px * 128 + ((s ^ ((px >> 1) & 7)) << 4)) = v[u];
      }
    }
  };

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  RD_W_LOAD(wA, 0);
  if (nsteps > 1) RD_W_LOAD(wB, 1);
  float* Sc = (float*)(Ws + NT * 32 * 128);  // [scale(Cout) | shift(Cout)] for the epilogue
  if (tid < NT * 32) {
    Sc[tid] = a.scale ? a.scale[tid] : 1.f;
    Sc[NT * 32 + tid] = a.shift ? a.shift[tid] : 0.f;
  }
  a_stage(0);

  // steps are processed in pairs so that the two register sets are addressed statically
  for (int step = 0; step < nsteps; ++step) {
    const int chunk = step / a.ntaps, tap = step - chunk * a.ntaps;
    __syncthreads();  // every wave is done reading the previous slab (and, at a chunk boundary, the halo tile)
    if (tap == 0 && chunk > 0 && !(a.dbg & 4)) a_stage(chunk);
    if (!(a.dbg & 2) || step == 0) {
      if (step & 1) {
        RD_W_STORE(wB);
        if (step + 2 < nsteps) RD_W_LOAD(wB, step + 2);
      } else {
        RD_W_STORE(wA);
        if (step + 2 < nsteps) RD_W_LOAD(wA, step + 2);
      }
    }
    __syncthreads();

    const int ns_c = min(8, a.nslots - 8 * chunk);
    const int tdh = (int)((a.dh_pack >> (4 * tap)) & 15) - 8, tdw = (int)((a.dw_pack >> (4 * tap)) & 15) - 8;
    const int trow = wm + (tdh - a.min_dh);
    int a_off[2], a_swz[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      int px = trow * a.CI + (mt * 32 + m) * a.in_stride + (tdw - a.min_dw);
      a_off[mt] = px * 128;
      a_swz[mt] = (px >> 1) & 7;
    }
    for (int ks = 0; ks < ((a.dbg & 8) ? 0 : (ns_c >> 1)); ++ks) {
      const int slot = 2 * ks + hi;
      if constexpr (DT == RD_BF16) {
        s16x8 av[2], bv[NT];
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = *(const s16x8*)(As + a_off[i] + ((slot ^ a_swz[i]) << 4));
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[j] = *(const s16x8*)(Ws + b_row[j] + ((slot ^ b_swz[j]) << 4));
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[j], av[i], acc[i][j], 0, 0, 0);
      } else {
        f32x4 av[2], bv[NT];
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = *(const f32x4*)(As + a_off[i] + ((slot ^ a_swz[i]) << 4));
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[j] = *(const f32x4*)(Ws + b_row[j] + ((slot ^ b_swz[j]) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j][e], av[i][e], acc[i][j], 0, 0, 0);
      }
    }
  }
#undef RD_W_LOAD
#undef RD_W_STORE

  // ---- epilogue: BN affine, ReLU / residual, store.  The MFMAs are issued "transposed" (A operand = weights,
  // B operand = pixels) and the weight rows are permuted in LDS, so lane (px, hi) holds output channels
  // 32*nt + 16*hi + r (r = 0..15) of its pixel: 16 contiguous channels -> 16-byte residual loads and stores.
  const int oh = h0 + wm;
  if (oh >= a.H) return;
  T* __restrict__ y = (T*)a.y + (size_t)b * a.y_bs + (size_t)oh * a.Wout * a.y_cs + a.y_co;
  const T* __restrict__ res = (const T*)a.res + (size_t)b * a.r_bs + (size_t)oh * a.Wout * a.r_cs + a.r_co;
  const bool relu_pre = a.flags & RD_RELU_PRE, do_add = a.flags & RD_ADD, relu_post = a.flags & RD_RELU_POST;
  constexpr int SPT = 16 / E::CH;                 // 16-byte slots per 16 channels (2 for bf16, 4 for f32)
  constexpr int NB = (DT == RD_BF16) ? NT : 1;    // tiles whose residual loads are batched ahead of the math
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int q = q0 + mt * 32 + m;
    const bool live = q < a.Wq;
    const size_t pix = (size_t)q * a.out_stride + a.out_off;
#pragma unroll
    for (int nb = 0; nb < NT; nb += NB) {
      Slot16 rv[NB][SPT];
      if (do_add && live) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int u = 0; u < SPT; ++u)
            rv[j][u] = *(const Slot16*)(res + pix * a.r_cs + (nb + j) * 32 + 16 * hi + u * E::CH);
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int nt = nb + j, cb = nt * 32 + 16 * hi;
        T rr[16];
        if (do_add && live) memcpy(rr, rv[j], sizeof(rr));
        T out[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 sc = *(const f32x4*)(Sc + cb + 4 * g);
          const f32x4 sh = *(const f32x4*)(Sc + NT * 32 + cb + 4 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            float v = acc[mt][nt][r] * sc[e] + sh[e];
            if (relu_pre) v = fmaxf(v, 0.f);
            if (do_add && live) v += E::to_f32(rr[r]);
            if (relu_post) v = fmaxf(v, 0.f);
            out[r] = E::from_f32(v);
          }
        }
        if (live && !(a.dbg & 1)) {
          Slot16 pk[SPT];
          memcpy(pk, out, sizeof(out));
#pragma unroll
          for (int u = 0; u < SPT; ++u) *(Slot16*)(y + pix * a.y_cs + cb + u * E::CH) = pk[u];
        }
      }
    }
  }
}

// ---- host side: tap lists, packing, launch -------------------------------------------------------------
struct TapList {
  int n = 0;
  int dh[9], dw[9], kh[9], kw[9];
};
inline TapList conv_taps(int kh, int kw) {  // pad = (k-1)/2, dilate 1   (mxnext/simple.py:131-135)
  TapList t;
  for (int i = 0; i < kh; ++i)
    for (int j = 0; j < kw; ++j) {
      t.kh[t.n] = i; t.kw[t.n] = j;
      t.dh[t.n] = i - (kh - 1) / 2;
      t.dw[t.n] = j - (kw - 1) / 2;
      ++t.n;
    }
  return t;
}
// transposed conv, stride (1,s), pad (1,pad_w), kernel (3,kw): output column ow = q*s + phase receives
// x[h + 1 - i][q + (phase + pad_w - j)/s] * Wd[ci][co][i][j] for the j with (phase + pad_w - j) % s == 0.
inline TapList deconv_taps(int kh, int kw, int s, int pad_w, int phase) {
  TapList t;
  const int pad_h = (kh - 1) / 2;
  for (int i = 0; i < kh; ++i)
    for (int j = 0; j < kw; ++j) {
      int num = phase + pad_w - j;
      if (((num % s) + s) % s != 0) continue;
      if (t.n >= 9) { t.n = 10; return t; }
      t.kh[t.n] = i; t.kw[t.n] = j;
      t.dh[t.n] = pad_h - i;
      t.dw[t.n] = num / s;  // exact
      ++t.n;
    }
  return t;
}
inline int cin_slots(int cin, int dt) {  // channels rounded up to one MFMA k-step (two 16-byte slots)
  int ch = ch_per_slot(dt);
  return round_up(cin, 2 * ch) / ch;
}
inline size_t conv_packed_bytes(int ntaps, int cin, int cout, int dt) {
  int nchunk = (cin_slots(cin, dt) + 7) / 8;
  return (size_t)nchunk * ntaps * cout * 128;
}
// get(co, ci, tap) -> float
template <class F>
inline void pack_taps(int ntaps, int cin, int cout, int dt, void* out, F get) {
  const int ch = ch_per_slot(dt), nchunk = (cin_slots(cin, dt) + 7) / 8, kc = 8 * ch;
  for (int c = 0; c < nchunk; ++c)
    for (int t = 0; t < ntaps; ++t)
      for (int co = 0; co < cout; ++co)
        for (int k = 0; k < kc; ++k) {
          int ci = c * kc + k;
          float v = ci < cin ? get(co, ci, t) : 0.f;
          size_t idx = (((size_t)c * ntaps + t) * cout + co) * kc + k;
          if (dt == RD_BF16) ((bf16_t*)out)[idx] = f32_to_bf16(v);
          else ((float*)out)[idx] = v;
        }
}

inline int launch_conv(const TapList& tl, const void* x, int x_cs, int x_co, const void* w, const float* scale,
                       const float* shift, const void* res, int r_cs, int r_co, void* y, int y_cs, int y_co,
                       int B, int H, int Win, int Wq, int Wout, int cin, int cout, int in_stride,
                       int out_stride, int out_off, int flags, int dt, hipStream_t st) {
  RD_REQUIRE(dt == RD_F32 || dt == RD_BF16, RD_EINVAL, "conv: dtype %d", dt);
  RD_REQUIRE(cout == 64 || cout == 128, RD_ESHAPE, "conv: cout %d not in {64,128}", cout);
  RD_REQUIRE(tl.n >= 1 && tl.n <= 9, RD_ESHAPE, "conv: %d taps unsupported", tl.n);
  RD_REQUIRE(B > 0 && H > 0 && Win > 0 && Wq > 0 && cin > 0, RD_ESHAPE, "conv: empty shape");
  const int ch = ch_per_slot(dt);
  RD_REQUIRE(x_cs % ch == 0 && x_co % ch == 0, RD_ESHAPE, "conv: x channel stride/offset must be 16-byte multiples");
  RD_REQUIRE(x_co + cin_slots(cin, dt) * ch <= x_cs, RD_ESHAPE, "conv: x buffer narrower than padded cin");
  RD_REQUIRE(!(flags & RD_ADD) || res, RD_EINVAL, "conv: RD_ADD without residual");
  RD_REQUIRE(y_cs % ch == 0 && y_co % ch == 0 && (!(flags & RD_ADD) || (r_cs % ch == 0 && r_co % ch == 0)), RD_ESHAPE,
             "conv: y / residual channel stride and offset must be 16-byte multiples");
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.x_cs = x_cs; a.x_co = x_co; a.x_bs = (long)H * Win * x_cs;
  a.w = w; a.scale = scale; a.shift = shift;
  a.res = res; a.r_cs = r_cs; a.r_co = r_co; a.r_bs = (long)H * Wout * r_cs;
  a.y = y; a.y_cs = y_cs; a.y_co = y_co; a.y_bs = (long)H * Wout * y_cs;
  a.H = H; a.Win = Win; a.Wq = Wq; a.Wout = Wout;
  a.nslots = cin_slots(cin, dt); a.cout = cout; a.ntaps = tl.n;
  a.in_stride = in_stride; a.out_stride = out_stride; a.out_off = out_off; a.flags = flags;
  { const char* e = getenv("RD_CONV_DBG"); a.dbg = e ? atoi(e) : 0; }
  int mndh = 99, mxdh = -99, mndw = 99, mxdw = -99;
  for (int t = 0; t < tl.n; ++t) {
    a.dh_pack |= (unsigned long long)(tl.dh[t] + 8) << (4 * t);
    a.dw_pack |= (unsigned long long)(tl.dw[t] + 8) << (4 * t);
    mndh = std::min(mndh, tl.dh[t]); mxdh = std::max(mxdh, tl.dh[t]);
    mndw = std::min(mndw, tl.dw[t]); mxdw = std::max(mxdw, tl.dw[t]);
  }
  const int RO = 4;
  a.min_dh = mndh; a.min_dw = mndw;
  a.RI = RO + (mxdh - mndh);
  a.CI = 63 * in_stride + (mxdw - mndw) + 1;
  const size_t lds = (size_t)a.RI * a.CI * 128 + (size_t)cout * 128 + (size_t)cout * 8;
  RD_REQUIRE(lds <= 160 * 1024, RD_ESHAPE, "conv: LDS tile %zu B too large", lds);
  dim3 grid((Wq + 63) / 64, (H + RO - 1) / RO, B);
  ProfScope ps(RD_PROF_CONV, st);
  if (dt == RD_BF16) {
    if (cout == 64) hipLaunchKernelGGL((conv_taps_kernel<RD_BF16, 2>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((conv_taps_kernel<RD_BF16, 4>), grid, dim3(256), lds, st, a);
  } else {
    if (cout == 64) hipLaunchKernelGGL((conv_taps_kernel<RD_F32, 2>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((conv_taps_kernel<RD_F32, 4>), grid, dim3(256), lds, st, a);
  }
  return check_launch("conv_taps_kernel");
}

}  // namespace rd
