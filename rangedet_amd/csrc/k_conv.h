// Generic implicit-GEMM convolution over channels-last range images for gfx950 (MFMA), with the BatchNorm affine,
// ReLU and residual add fused into the epilogue.  One kernel serves every conv-shaped layer of the RangeDet
// graph through a *tap list*:  y[h, q*out_stride+out_off, co] = sum_t sum_ci x[h+dh_t, q*in_stride+dw_t, ci] * W_t[ci][co]
//   3x3 / 1x1 convs, stride (1,1) or (1,2)        dla_backbone.py:18-56, head/builder.py:221-240 (reference)
//   transposed convs k(3,8)s(1,4) / k(3,4)s(1,2)  dla_backbone.py:117-127  -> one launch per output phase
// It runs the whole graph in fp32 (the parity mode) and, in bf16, what the persistent 3x3 kernel (k_conv3.h) does not
// take: the 1x1 convs.
//
// Work decomposition (wave64, 256-thread workgroups, two per CU):
//   workgroup tile = 4 output rows x 64 output columns x 64 output channels (Cout = 128: two channel-half workgroups
//   per pixel tile, adjacent in the XCD-aware tile order); each wave owns one row = two 32-px MFMA tiles x two 32-channel
//   tiles -> 4 accumulators of 32x32.
//   The input halo tile (6 rows x 66 columns x one 128-byte k-chunk for a 3x3) is staged ONCE per k-chunk in LDS
//   (global -> registers -> LDS) and re-read by all taps; the per-(chunk,tap) weight slab [64 Cout][128 B] arrives by
//   LDS-DMA into a 3-deep ring (8-deep for launches with few workgroups) with counted s_waitcnt vmcnt(N) and a raw
//   s_barrier per step -- see the comments at RD_LDS_BARRIER for why not __syncthreads().
//   LDS rows are 128 B (= 8 slots of 16 B = 64 bf16 / 32 f32 channels); slot index is XOR-swizzled with
//   ((row>>1)&7) so the 16-lane groups of ds_read_b128 hit 16 distinct slots (conflict-free for stride-1 pixels).
//   MFMA operands: lanes 0-31 read slot 2*ks, lanes 32-63 slot 2*ks+1 of their pixel / output channel:
//     bf16: that IS the v_mfma_f32_32x32x16_bf16 fragment (k = 8*(lane>>5)+j);
//     f32 : 4 x v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain), element e of the slot pairs k = 4*(2ks+hi)+e.
//   bf16 weights come packed in MFMA-fragment order (k_conv3.h pack_taps_frag); the slab fill gathers from it.
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
  int ncol;  // column tiles (64 output positions each)
  unsigned ci_magic;  // floor(2^32 / CI) + 1: px / CI == umulhi(px, ci_magic) for the px range of a halo tile
  unsigned long long* trace;  // dev tracing (tools/conv_trace.py): 8 timestamps per workgroup, or null
};

// weight row held by A-operand lane row mm, so that the D registers of a lane are 16 consecutive output channels
__host__ __device__ inline int conv_row_perm(int mm) { return 16 * ((mm >> 2) & 1) + 4 * (mm >> 3) + (mm & 3); }

// s_waitcnt immediate (gfx9 encoding) that waits for vmcnt <= n only (expcnt / lgkmcnt left at "no wait")
#define RD_VMCNT_IMM(n) (((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))

template <int DT, int NW, int RING>  // NW waves (= output rows) per workgroup, RING weight slabs in LDS
__global__ __launch_bounds__(NW * 64) void conv_taps_kernel(ConvArgs a) {
  using E = Elem<DT>;
  using T = typename E::T;
  HIP_DYNAMIC_SHARED(unsigned char, smem);
  constexpr int NTH = NW * 64;
  constexpr int COUT = 64;                // output channels per workgroup; Cout = 128 runs as two channel-half workgroups
  constexpr int RO = NW;                  // output rows per workgroup, one wave each
  constexpr int SLAB = COUT * 128;        // bytes of this workgroup's half of one (k-chunk, tap) weight slab
  constexpr int IPW = COUT / 8 / NW;      // LDS-DMA instructions (64 x 16 B) per wave per slab
  constexpr int D = RING - 1;             // prefetch distance in steps
  static_assert(IPW * NW * 64 * 16 == SLAB && IPW >= 1, "slab must split evenly over the waves");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave;
  const int nhalf = a.cout >> 6;          // 1 or 2 channel halves
  const int wc = 0;
  // XCD-aware tile assignment.  Workgroup ids are dealt round-robin to the 8 XCDs (private L2s), so each XCD gets a
  // CONTIGUOUS range of the tile list ordered (row block, column tile, channel half): the two channel halves of a
  // pixel tile run back to back on one XCD (second one hits the halo in L2) and vertically adjacent row blocks, which
  // share halo rows, live in the same L2.  Pure speed choice; any mapping is correct.
  int widx;
  {
    const int T = gridDim.x, L = blockIdx.x, xcd = L & 7, i = L >> 3;
    const int qn = T >> 3, rn = T & 7;
    widx = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + i;
  }
  const int chalf = widx % nhalf;
  const int ct = (widx / nhalf) % a.ncol;
  const int q0 = ct * 64, h0 = (widx / (nhalf * a.ncol)) * RO, b = blockIdx.z;
  const int m = lane & 31, hi = lane >> 5;
  int tpt = 0;
#define RD_TRACE() { if (a.trace && tid == 0 && tpt < 8) a.trace[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8 + tpt++] = wall_clock64(); }
  RD_TRACE()

  unsigned char* As = smem;
  unsigned char* Ws = smem + a.RI * a.CI * 128;            // RING slabs
  float* Sc = (float*)(Ws + RING * SLAB);                  // [scale(Cout) | shift(Cout)]
  const T* x = (const T*)a.x + (size_t)b * a.x_bs;
  const int nchunk = (a.nslots + 7) >> 3;
  const int nsteps = nchunk * a.ntaps;

  // weight slabs go global -> LDS by LDS-DMA: the destination of one instruction is wave-uniform base + lane*16,
  // so the LDS image is linear and both the XOR swizzle and the output-channel row permutation are applied to the
  // per-lane SOURCE address (constant across steps).
  int goff[IPW];
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int p = (wave * IPW + j) * 64 + lane;            // linear 16-byte slot of the LDS slab
    const int n = p >> 3, sp = p & 7;
    const int s = sp ^ ((n >> 1) & 7);
    const int mm = n & 31;
    if constexpr (DT != RD_F32) {
      // bf16 weights are packed in MFMA-fragment order (k_conv3.h pack_taps_frag): [32-ch chunk][tap][ks][Cout/32][lane]
      const int ncb = a.cout >> 5, cb = chalf * 2 + (n >> 5);
      goff[j] = (((s >> 2) * a.ntaps * 2 + ((s >> 1) & 1)) * ncb + cb) * 1024 + ((s & 1) * 32 + mm) * 16;
    } else {
      const int c = (n & ~31) | conv_row_perm(mm);
      goff[j] = (c * 8 + s) * 16;
    }
  }
  int fch = 0, ftap = 0;                                   // (64-ch chunk, tap) of the next slab to fetch
  auto w_fill = [&](int step) {
    const unsigned char* src;
    if constexpr (DT != RD_F32) {
      src = (const unsigned char*)a.w + (size_t)((2 * fch * a.ntaps + ftap) * 2 * (a.cout >> 5)) * 1024;
      if (++ftap == a.ntaps) { ftap = 0; ++fch; }
    } else {
      src = (const unsigned char*)a.w + ((size_t)step * nhalf + chalf) * SLAB;
    }
    unsigned char* dst = Ws + (step % RING) * SLAB + wave * IPW * 1024;
#pragma unroll
    for (int j = 0; j < IPW; ++j)
      lds_dma16(src + goff[j], dst + j * 1024);
  };
  // halo tile: all global loads of a pass are issued before the first LDS write (latency paid once per pass)
  constexpr int AP = (6 * 66 * 8 + NTH - 1) / NTH > 8 ? 8 : (6 * 66 * 8 + NTH - 1) / NTH;
  auto a_stage = [&](int chunk) {
    const int ns_c = min(8, a.nslots - 8 * chunk);
    const int items = a.RI * a.CI * 8;
    for (int base = 0; base < items; base += NTH * AP) {
      Slot16 v[AP];
#pragma unroll
      for (int u = 0; u < AP; ++u) {
        const int idx = base + u * NTH + tid;
        const int px = idx >> 3, s = idx & 7;
        const int r = (int)__umulhi((unsigned)px, a.ci_magic), cc = px - r * a.CI;  // px / CI without a divide
        const int ih = h0 + a.min_dh + r, iw = q0 * a.in_stride + a.min_dw + cc;
        v[u] = Slot16{0u, 0u, 0u, 0u};
        if (idx < items && s < ns_c && ih >= 0 && ih < a.H && iw >= 0 && iw < a.Win)
          v[u] = *(const Slot16*)(x + ((size_t)ih * a.Win + iw) * a.x_cs + a.x_co + (chunk * 8 + s) * E::CH);
      }
#pragma unroll
      for (int u = 0; u < AP; ++u) {
        const int idx = base + u * NTH + tid;
        const int px = idx >> 3, s = idx & 7;
        if (idx < items) *(Slot16*)(As + px * 128 + ((s ^ ((px >> 1) & 7)) << 4)) = v[u];
      }
    }
    // tell hipcc's waitcnt pass explicitly that nothing of this stage is pending any more: the loads above sit in
    // predicated blocks, and without this it keeps a "maybe pending" state on their registers that costs an
    // s_waitcnt vmcnt(0) (= a drained weight ring) in every iteration of the tap loop.
    __builtin_amdgcn_s_waitcnt(RD_VMCNT_IMM(0));
  };
  // Workgroup barrier that orders LDS traffic only.  __syncthreads() (and a workgroup fence, even one restricted to
  // the "local" address space on this compiler) makes hipcc emit s_waitcnt vmcnt(0), which would drain the weight
  // ring at every step; the hardware only needs this wave's LDS accesses retired (lgkmcnt(0)) before s_barrier.
  // The empty asm statements stop the compiler from moving LDS accesses across the barrier.
#define RD_LDS_BARRIER()                                                     \
  {                                                                          \
    asm volatile("" ::: "memory");                                           \
    __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0), vmcnt/expcnt free */   \
    __builtin_amdgcn_s_barrier();                                            \
    asm volatile("" ::: "memory");                                           \
  }

  // Fragment addresses: a 128-byte LDS row holds 8 XOR-swizzled 16-byte slots; lane (row, hi) reads slot 2*ks + hi, i.e.
  // byte (row*128) | (((swz ^ hi) ^ 2*ks) << 4) = addr0 ^ (ks << 5): one v_xor per fragment read in the MFMA loop.
  int b0[2], pbase[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = wc * 64 + nt * 32 + m;
    b0[nt] = (n << 7) | (((((n >> 1) & 7)) ^ hi) << 4);
    pbase[nt] = wm * a.CI + (nt * 32 + m) * a.in_stride;   // halo pixel of this lane for tap (min_dh, min_dw)
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  a_stage(0);                                   // ordinary loads first: hipcc waits vmcnt(0) for them, nothing else in flight yet
  if (tid < COUT) {
    Sc[tid] = a.scale ? a.scale[chalf * 64 + tid] : 1.f;
    Sc[COUT + tid] = a.shift ? a.shift[chalf * 64 + tid] : 0.f;
  }
#pragma unroll
  for (int s0 = 0; s0 < D; ++s0)
    if (s0 < nsteps) w_fill(s0);
  RD_TRACE()

  // Loop nest: k-chunks outside (halo restage = ordinary loads), taps inside (LDS-DMA only).  Keeping the ordinary
  // loads out of the tap loop matters: hipcc's waitcnt pass otherwise carries "maybe pending load" state of the halo
  // registers around the loop and drains vmcnt(0) in every iteration.
  int step = 0;
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    if (chunk > 0) {
      RD_TRACE()
      RD_LDS_BARRIER();  // every wave is done with the previous chunk's halo tile
      a_stage(chunk);
      RD_TRACE()
    }
   for (int tap = 0; tap < a.ntaps; ++tap, ++step) {
    // slab `step` must have landed: at most the younger fills (steps step+1 .. step+D-1) may still be in flight
    {
      const int younger = min(D - 1, nsteps - 1 - step);
      if (younger >= 7) __builtin_amdgcn_s_waitcnt(RD_VMCNT_IMM(7 * IPW));
      else if (younger == 6) __builtin_amdgcn_s_waitcnt(RD_VMCNT_IMM(6 * IPW));
      else if (younger == 5) __builtin_amdgcn_s_waitcnt(RD_VMCNT_IMM(5 * IPW));
      else if (younger == 4) __builtin_amdgcn_s_waitcnt(RD_VMCNT_IMM(4 * IPW));
      else if (younger == 3) __builtin_amdgcn_s_waitcnt(RD_VMCNT_IMM(3 * IPW));
      else if (younger == 2) __builtin_amdgcn_s_waitcnt(RD_VMCNT_IMM(2 * IPW));
      else if (younger == 1) __builtin_amdgcn_s_waitcnt(RD_VMCNT_IMM(1 * IPW));
      else __builtin_amdgcn_s_waitcnt(RD_VMCNT_IMM(0));
    }
    RD_LDS_BARRIER();  // all parts of slab `step` (and a fresh halo tile) are in LDS; everyone is done with slab step-1
    if (step + D < nsteps) w_fill(step + D);   // refill the slot slab step-1 just vacated

    const int ns_c = min(8, a.nslots - 8 * chunk);
    const int tdh = (int)((a.dh_pack >> (4 * tap)) & 15) - 8, tdw = (int)((a.dw_pack >> (4 * tap)) & 15) - 8;
    const int delta = (tdh - a.min_dh) * a.CI + (tdw - a.min_dw);   // wave-uniform halo-pixel shift of this tap
    int a0[2], bw[2];
    const int slab_off = (a.RI * a.CI + (step % RING) * (SLAB / 128)) << 7;  // ring slot, bytes from smem
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int px = pbase[i] + delta;
      a0[i] = (px << 7) | ((((px >> 1) & 7) ^ hi) << 4);
      bw[i] = b0[i] + slab_off;
    }
    auto kstep = [&](int ks) {
      const int kx = ks << 5;
      if constexpr (DT != RD_F32) {
        s16x8 av[2], bv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          av[i] = *(const s16x8*)(smem + (a0[i] ^ kx));
          bv[i] = *(const s16x8*)(smem + (bw[i] ^ kx));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i][j] = H16<DT>::mfma(bv[j], av[i], acc[i][j]);
      } else {
        f32x4 av[2], bv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          av[i] = *(const f32x4*)(smem + (a0[i] ^ kx));
          bv[i] = *(const f32x4*)(smem + (bw[i] ^ kx));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j][e], av[i][e], acc[i][j], 0, 0, 0);
      }
    };
    for (int ks = 0; ks < (ns_c >> 1); ++ks) kstep(ks);
   }
  }
#undef RD_LDS_BARRIER
  RD_TRACE()

  // ---- epilogue: BN affine, ReLU / residual, store.  The MFMAs are issued "transposed" (A operand = weights,
  // B operand = pixels) and the weight rows are permuted in LDS, so lane (px, hi) holds output channels
  // 64*wc + 32*nt + 16*hi + r (r = 0..15) of its pixel: 16 contiguous channels -> 16-byte residual loads and stores.
  const int oh = h0 + wm;
  if (oh >= a.H) { RD_TRACE() return; }
  T* __restrict__ y = (T*)a.y + (size_t)b * a.y_bs + (size_t)oh * a.Wout * a.y_cs + a.y_co + chalf * 64;
  const T* __restrict__ res = (const T*)a.res + (size_t)b * a.r_bs + (size_t)oh * a.Wout * a.r_cs + a.r_co + chalf * 64;
  const bool relu_pre = a.flags & RD_RELU_PRE, do_add = a.flags & RD_ADD, relu_post = a.flags & RD_RELU_POST;
  constexpr int SPT = 16 / E::CH;                 // 16-byte slots per 16 channels (2 for bf16, 4 for f32)
  constexpr int NB = (DT != RD_F32) ? 2 : 1;     // tiles whose residual loads are batched ahead of the math
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int q = q0 + mt * 32 + m;
    const bool live = q < a.Wq;
    const size_t pix = (size_t)q * a.out_stride + a.out_off;
#pragma unroll
    for (int nb = 0; nb < 2; nb += NB) {
      Slot16 rv[NB][SPT];
      if (do_add && live) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int u = 0; u < SPT; ++u)
            rv[j][u] = *(const Slot16*)(res + pix * a.r_cs + wc * 64 + (nb + j) * 32 + 16 * hi + u * E::CH);
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int nt = nb + j, cb = wc * 64 + nt * 32 + 16 * hi;
        T rr[16];
        if (do_add && live) memcpy(rr, rv[j], sizeof(rr));
        T out[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 sc = *(const f32x4*)(Sc + cb + 4 * g);
          const f32x4 sh = *(const f32x4*)(Sc + COUT + cb + 4 * g);
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
        if (live) {
          Slot16 pk[SPT];
          memcpy(pk, out, sizeof(out));
#pragma unroll
          for (int u = 0; u < SPT; ++u) *(Slot16*)(y + pix * a.y_cs + cb + u * E::CH) = pk[u];
        }
      }
    }
  }
  RD_TRACE()
#undef RD_TRACE
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
// Every packed conv-family weight image ends in RD_CONV_TAIL zero bytes (written by the packers): the persistent 3x3 kernel
// uses them as the DMA source of padding pixels / channels, so the library needs no device allocation of its own.
constexpr size_t RD_CONV_TAIL = 256;
inline size_t conv_packed_body_bytes(int ntaps, int cin, int cout, int dt) {
  int nchunk = (cin_slots(cin, dt) + 7) / 8;
  return (size_t)nchunk * ntaps * cout * 128;
}
inline size_t conv_packed_bytes(int ntaps, int cin, int cout, int dt) {
  return conv_packed_body_bytes(ntaps, cin, cout, dt) + RD_CONV_TAIL;
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
          if (is_h16(dt)) ((bf16_t*)out)[idx] = h16_from_f32(dt, v);
          else ((float*)out)[idx] = v;
        }
}

// dev tracing (tools/conv_trace.py): the CALLER registers a device buffer of CONV_TRACE_CAP 64-bit words (rd_dev_conv_trace_set) that
// the conv kernels stamp with s_memrealtime (100 MHz) per phase; nullptr = off.  The library itself never allocates device memory.
constexpr size_t CONV_TRACE_CAP = 1 << 20;
inline unsigned long long*& conv_trace_slot() {
  static unsigned long long* buf = nullptr;
  return buf;
}
inline unsigned long long* conv_trace_buf() { return conv_trace_slot(); }

// k_conv1.h: streaming 1x1 bf16 kernel
inline bool conv1_eligible(const TapList& tl, int in_stride, int out_stride, int cin, int cout, int dt, int Win, int Wq, int Wout);
inline int launch_conv1(const void* x, int x_cs, int x_co, const void* w, const float* scale, const float* shift,
                        const void* res, int r_cs, int r_co, void* y, int y_cs, int y_co, int B, int H, int Win, int Wout,
                        int cin, int cout, int flags, int sw, hipStream_t st);
// k_conv3.h: persistent 3x3 stride-1 bf16 kernel
inline bool conv3_eligible(const TapList& tl, int in_stride, int out_stride, int cout, int dt, int Win, int Wq, int Wout);
inline int launch_conv3(const void* x, int x_cs, int x_co, const void* w, const float* scale, const float* shift,
                        const void* res, int r_cs, int r_co, void* y, int y_cs, int y_co, int B, int H, int W, int cin,
                        int cout, int flags, int sw, hipStream_t st, int ts, const struct Conv3Args* head, int dt,
                        const struct Conv3Second* g1 = nullptr, const struct Conv3Phases* ph = nullptr,
                        const struct Conv3Src2* s2 = nullptr, int body = 0);

inline int launch_conv(const TapList& tl, const void* x, int x_cs, int x_co, const void* w, const float* scale,
                       const float* shift, const void* res, int r_cs, int r_co, void* y, int y_cs, int y_co,
                       int B, int H, int Win, int Wq, int Wout, int cin, int cout, int in_stride,
                       int out_stride, int out_off, int flags, int dt, hipStream_t st) {
  RD_REQUIRE(dt == RD_F32 || is_h16(dt), RD_EINVAL, "conv: dtype %d", dt);
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
  int mndh = 99, mxdh = -99, mndw = 99, mxdw = -99;
  for (int t = 0; t < tl.n; ++t) {
    a.dh_pack |= (unsigned long long)(tl.dh[t] + 8) << (4 * t);
    a.dw_pack |= (unsigned long long)(tl.dw[t] + 8) << (4 * t);
    mndh = std::min(mndh, tl.dh[t]); mxdh = std::max(mxdh, tl.dh[t]);
    mndw = std::min(mndw, tl.dw[t]); mxdw = std::max(mxdw, tl.dw[t]);
  }
  if (conv1_eligible(tl, in_stride, out_stride, cin, cout, dt, Win, Wq, Wout) && x_co + 16 * ((cin + 15) / 16) <= x_cs)
    return launch_conv1(x, x_cs, x_co, w, scale, shift, res, r_cs, r_co, y, y_cs, y_co, B, H, Win, Wout, cin, cout, flags, in_stride, st);
  if (conv3_eligible(tl, in_stride, out_stride, cout, dt, Win, Wq, Wout))
    return launch_conv3(x, x_cs, x_co, w, scale, shift, res, r_cs, r_co, y, y_cs, y_co, B, H, Win, cin, cout, flags, in_stride, st, 0, nullptr, dt);
  // Workgroup = 4 rows x 64 px x 64 output channels (Cout = 128 runs as two channel-half workgroups per pixel tile),
  // 4 waves, 3-deep weight ring, two workgroups per CU when the halo allows.
  constexpr int RO = 4;
  const int mxw = (mxdw - mndw);
  a.min_dh = mndh; a.min_dw = mndw;
  a.RI = RO + (mxdh - mndh);
  a.CI = 63 * in_stride + mxw + 1;
  const size_t halo = (size_t)a.RI * a.CI * 128;
  const int nblocks = ((Wq + 63) / 64) * ((H + RO - 1) / RO) * (cout / 64);
  // few workgroups (low-resolution layers): at most one per CU anyway, so spend the idle LDS on a deeper weight ring --
  // such layers are a pure latency chain of (k-chunk, tap) steps and run at ring depth / L2 latency.
  const bool deep = nblocks * B <= 320 && halo + 8 * 64 * 128 + 512 <= 160 * 1024;
  const int ring = deep ? 8 : 3;
  const size_t lds = halo + (size_t)ring * 64 * 128 + 64 * 8;
  RD_REQUIRE(lds <= 160 * 1024, RD_ESHAPE, "conv: LDS tile %zu B too large", lds);
  a.ncol = (Wq + 63) / 64;
  a.ci_magic = (unsigned)((1ull << 32) / (unsigned)a.CI) + 1u;
  dim3 grid(a.ncol * ((H + RO - 1) / RO) * (cout / 64), 1, B);
  if (conv_trace_buf() && (size_t)grid.x * B * 8 <= CONV_TRACE_CAP) a.trace = conv_trace_buf();
  ProfScope ps(RD_PROF_CONV, st);
#define RD_LAUNCH_CONV(DT_)                                                                            \
  if (deep) hipLaunchKernelGGL((conv_taps_kernel<DT_, 4, 8>), grid, dim3(256), lds, st, a);             \
  else hipLaunchKernelGGL((conv_taps_kernel<DT_, 4, 3>), grid, dim3(256), lds, st, a);
  if (dt == RD_BF16) { RD_LAUNCH_CONV(RD_BF16) } else if (dt == RD_F16) { RD_LAUNCH_CONV(RD_F16) } else { RD_LAUNCH_CONV(RD_F32) }
#undef RD_LAUNCH_CONV
  return check_launch("conv_taps_kernel");
}

}  // namespace rd
