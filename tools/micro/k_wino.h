// EXPERIMENT (round 6): 3x3 / stride-1 / pad-1 16-bit convolution, cin = k*32 -> cout = 128, as 1-D Winograd F(2,3)
// along the azimuth axis W on the matrix cores -- 12 instead of 18 tap-GEMMs per output pixel PAIR (2/3 of the direct form's MFMAs).
// Reference layers: the 128 -> 128 tower convs (head/builder.py:221-240) and BasicBlock convs (dla_backbone.py:18-56).
//
//   y[h][2p + e][co] = sum_dh sum_ci sum_dw x[h + dh - 1][2p + e + dw - 1][ci] * g[co][ci][dh][dw]          (direct)
//   V_m[h][p][ci] = B^T d, d = x[h][2p - 1 .. 2p + 2][ci]:  V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3' = d3 - d1 (= -V3)
//   U_m[co][ci][dh] = G g:   U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3' = -g2            (packed on the host)
//   M_m[h][p][co] = sum_dh sum_ci U_m[co][ci][dh] * V_m[h + dh - 1][p][ci]                                   (3 x 4 GEMMs on MFMA)
//   y[h][2p] = M0 + M1 + M2,  y[h][2p + 1] = M1 - M2 - M3                                                    (registers, fp32)
// V is rounded to the 16-bit type (one rounding more than the direct form on the input side), U is rounded once like any weight.
//
// One workgroup of 4 waves per CU (one wave per SIMD, 512 registers: 256 accumulators).  Tile = 4 output rows x 64 columns (32
// pairs) x 128 cout.  Wave w: row pair rp = w >> 1 (output rows 2rp, 2rp + 1), cout half ch = w & 1 -> acc[m 4][row 2][cf 2].
//   unit  = (tile, 32-channel chunk): raw halo 6 rows x 66 px x 64 B = 25 one-KB pieces, double buffered, fetched by LDS-DMA two units
//           ahead; pixel f = 66 r + c of a piece sits at 16-B position 4*pi(f & 15) + (slot ^ ((f >> 3) & 3)), pi(x) = (x >> 1) | (x & 1) << 3,
//           so that the transform's stride-2-pixel ds_read_b128 is conflict free;
//   V     = [ks half 2][m 4][row 6] fragments of 1 KB (lane (p, hi) = pair p, channels 16 ks + 8 hi ..): 2 x 24 KB.  Half ks is consumed by
//           the four steps (ks, m = 0..3) and rewritten for the NEXT unit while the other half is consumed;
//   step  = (unit, ks, m): one weight slab [dh 3][cf 4][64 lanes][8] = 12 KB in a 4-deep LDS ring (LDS-DMA, counted waits),
//           12 MFMAs per wave (dh x row x cf), 4 V + 6 U fragment reads; ONE workgroup barrier per step;
//   transform: per 4-step window 12 half-jobs (row 6 x {V0 V1 | V2 V3'}) = 3 per wave, one in each of the window's first three steps:
//           lane (p, hi) reads three raw pixels, writes two V slots (48 vector instructions per half-job).
#pragma once
#include "../../rangedet_amd/csrc/k_conv.h"

namespace rd {

constexpr int WN_TH = 4, WN_TW = 64, WN_PITCH = 66, WN_ROWS = 6, WN_NPX = WN_ROWS * WN_PITCH, WN_NPIECE = 25, WN_PPW = 7;
constexpr int WN_RAWB = WN_NPIECE * 1024, WN_VHALF = 4 * WN_ROWS * 1024, WN_SLAB = 3 * 4 * 1024, WN_R = 4;
constexpr int WN_V0 = 2 * WN_RAWB, WN_RING = WN_V0 + 2 * WN_VHALF;
constexpr size_t WN_LDS = WN_RING + WN_R * WN_SLAB;   // 51 200 + 49 152 + 49 152 = 149 504 B

struct WinoArgs {
  const bf16_t* x; int x_cs, x_co; long x_bs;
  const unsigned char* w;      // pack_wino_frag image
  const float* shift;
  const bf16_t* res; int r_cs, r_co; long r_bs;
  bf16_t* y; int y_cs, y_co; long y_bs;
  const unsigned char* zero16;
  int H, W, B, nchunk, flags, ncol, nrow, ntiles;
};

inline size_t wino_packed_body_bytes(int cin) { return (size_t)((cin + 31) / 32) * 8 * WN_SLAB; }
inline size_t wino_packed_bytes(int cin) { return wino_packed_body_bytes(cin) + RD_CONV_TAIL; }
// packed Winograd weights: [32-ch chunk][ks 2][m 4][dh 3][cout/32 = 4][64 lanes][8]; lane (mm, hi) of a fragment holds
// U_m[co = 32*cb + conv_row_perm(mm)][ci = 32*chunk + 16*ks + 8*hi + j][dh].   get(co, ci, dh, dw) -> float (BatchNorm scale folded in)
template <class F>
inline void pack_wino_frag(int cin, void* out, F get, int dt) {
  const int nchunk = (cin + 31) / 32;
  bf16_t* o = (bf16_t*)out;
  for (int c = 0; c < nchunk; ++c)
    for (int ks = 0; ks < 2; ++ks)
      for (int mi = 0; mi < 4; ++mi)
        for (int dh = 0; dh < 3; ++dh)
          for (int cb = 0; cb < 4; ++cb)
            for (int lane = 0; lane < 64; ++lane) {
              const int co = cb * 32 + conv_row_perm(lane & 31);
              for (int j = 0; j < 8; ++j) {
                const int ci = c * 32 + ks * 16 + (lane >> 5) * 8 + j;
                float u = 0.f;
                if (ci < cin) {
                  const float g0 = get(co, ci, dh, 0), g1 = get(co, ci, dh, 1), g2 = get(co, ci, dh, 2);
                  u = mi == 0 ? g0 : mi == 1 ? 0.5f * ((g0 + g2) + g1) : mi == 2 ? 0.5f * ((g0 + g2) - g1) : -g2;
                }
                *o++ = h16_from_f32(dt, u);
              }
            }
  memset(o, 0, RD_CONV_TAIL);
}

#define WN_WAIT_IMM(vm, lgkm) (((vm) & 15) | (((vm) >> 4) << 14) | (7 << 4) | ((lgkm) << 8))
// raw pieces a wave issues in step ordinal o of a unit (the window ks = 1 fetches the raw halo of unit u + 2)
constexpr int wn_nraw(int o) { return o < 4 ? 0 : o < 7 ? 2 : 1; }
constexpr int wn_raw_first(int o) { return o < 4 ? 0 : 2 * (o - 4); }
// DMA instructions a wave has issued after its part of slab g + 2 (issued last in step g - 2), as seen at the barrier of step g
constexpr int wn_younger(int o) { return 6 + wn_nraw((o + 7) & 7) + wn_nraw(o); }

template <int DT, int DBG = 0>
__global__ __launch_bounds__(256, 1) void wino3x3_stream_kernel(WinoArgs a) {
  HIP_DYNAMIC_SHARED(unsigned char, smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, hi = lane >> 5;
  const int rp = wave >> 1, ch = wave & 1;
  const int G = gridDim.x, wg = blockIdx.x;
  const int ntl = (a.ntiles - wg + G - 1) / G;            // tiles wg, wg + G, ... of this workgroup (grid <= ntiles)
  const int tiles_img = a.ncol * a.nrow;

#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem);
#endif
  auto dma_s = [&](const unsigned char* sbase, unsigned voff, int lds_off) {   // uniform base + 32-bit lane offset
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds0 + (unsigned)lds_off)) : "memory");
#else
    __builtin_amdgcn_global_load_lds(sbase + voff, smem + lds_off, 16, 0, 0);
#endif
  };
  auto dma_v = [&](const void* vptr, int lds_off) {                            // per-lane 64-bit address
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(vptr), "s"(__builtin_amdgcn_readfirstlane(lds0 + (unsigned)lds_off)) : "memory");
#else
    __builtin_amdgcn_global_load_lds(vptr, smem + lds_off, 16, 0, 0);
#endif
  };
#define WN_FENCE() __builtin_amdgcn_sched_barrier(0)

  // ---- raw halo fetch (two units ahead of the MFMAs) ----------------------------------------------------------------------------
  auto decode = [&](int t, int& ct, int& rb, int& b) {
    b = t / tiles_img;
    const int r_ = t - b * tiles_img;
    rb = r_ / a.ncol;
    ct = r_ - rb * a.ncol;
  };
  const int pxl = (((lane >> 2) & 7) << 1) | (lane >> 5);   // pixel (within its 16-pixel piece) whose slot this lane fetches
  const int ps = lane & 3;
  int hk = 0, hc = 0;                        // (tile ordinal, chunk) of the NEXT unit to fetch
  int hh0 = 0, hw0 = 0;                      // image coordinates of halo pixel (0, 0) of the fetch tile
  const unsigned char* htile = nullptr;      // uniform: halo pixel (0, 0) of the fetch tile, channel 0 (may lie outside the buffer)
  const unsigned char* hbase = nullptr;      // the same, chunk of the unit being fetched
  bool hnew = true;
  auto halo_begin = [&]() {
    if (hnew) {
      int ct, rb, b;
      decode(wg + hk * G, ct, rb, b);
      hh0 = rb * WN_TH - 1; hw0 = ct * WN_TW - 1;
      htile = (const unsigned char*)(a.x + (size_t)b * a.x_bs + a.x_co) + ((long)hh0 * a.W + hw0) * (long)a.x_cs * 2;
      hnew = false;
    }
    hbase = htile + hc * 64;
    if (hc + 1 < a.nchunk) ++hc;
    else if (hk + 1 < ntl) { ++hk; hc = 0; hnew = true; }   // past the end of the list: re-fetch the last unit (constant DMA counts)
  };
  auto halo_piece = [&](int buf, int j) {
    int q = wave * WN_PPW + j;
    q = q < WN_NPIECE ? q : WN_NPIECE - 1;               // (28 issue slots for 25 pieces: the last one is fetched up to four times)
    int ol = lane;
    asm volatile("" : "+v"(ol));                         // (keeps the per-piece address math out of the registers live across the MFMAs)
    const int px_ = (((ol >> 2) & 7) << 1) | (ol >> 5);
    const int f = 16 * q + px_, r = (f * 993) >> 16, c = f - WN_PITCH * r;
    const int s = (ol & 3) ^ ((f >> 3) & 3);
    const bool ok = f < WN_NPX && (unsigned)(hh0 + r) < (unsigned)a.H && (unsigned)(hw0 + c) < (unsigned)a.W && !(DBG & 16);
    const unsigned off = (unsigned)((r * a.W + c) * a.x_cs) * 2u + (unsigned)s * 16u;   // (>= 0 and < 2^31 within an image; 32-bit math)
    const unsigned char* src = hbase + off;
    dma_v(ok ? (const void*)src : (const void*)a.zero16, buf + q * 1024);
  };
  (void)pxl; (void)ps;

  // ---- weight slabs ---------------------------------------------------------------------------------------------------------------
  const int slab_bytes_tile = a.nchunk * 8 * WN_SLAB;
  int fsoff = 0;                                           // byte offset of the NEXT slab to fetch within the packed image
  auto slab_piece = [&](int slot, int j) {
    dma_s(a.w + (size_t)fsoff + (wave * 3 + j) * 1024, lane * 16, WN_RING + slot * WN_SLAB + (wave * 3 + j) * 1024);
  };
  auto slab_advance = [&]() { fsoff += WN_SLAB; fsoff = fsoff == slab_bytes_tile ? 0 : fsoff; };

  // ---- input transform: half-job i (0..2) of a window = row 2i + (wave >> 1), part = wave & 1 ------------------------------------
  // part 0: (a, b, c) = (d0, d1, d2) -> X = a - c = V0, Y = b + c = V1;  part 1: (a, b, c) = (d3, d2, d1) -> X = a - c = V3', Y = b - c = V2
  const int part = wave & 1;
  int ja[3], jb[3], jc[3];                                // LDS byte offsets (within a raw buffer, ks = 0) of the three pixels
  {
    auto raddr = [&](int f) { return ((f >> 4) << 10) + ((4 * (((f & 15) >> 1) | ((f & 1) << 3)) + (hi ^ ((f >> 3) & 3))) << 4); };
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int f0 = WN_PITCH * (2 * i + (wave >> 1)) + 2 * m + part;
      ja[i] = raddr(part ? f0 + 2 : f0); jb[i] = raddr(f0 + 1); jc[i] = raddr(part ? f0 : f0 + 2);
    }
  }
  const float jsgn = part ? -1.f : 1.f;
  // (the LDS bases below go through an opaque copy: folded into the per-use constants they exceed the 16-bit DS offset field and hipcc
  //  materialises every distinct base + constant in a register of its own -- dozens of them, spilled across the MFMA phase)
  int jwx = WN_V0 + lane * 16 + (wave >> 1) * 1024 + (part ? 3 : 0) * (WN_ROWS * 1024);   // V slot of X / Y (+ 2i rows, + half)
  int jwy = WN_V0 + lane * 16 + (wave >> 1) * 1024 + (part ? 2 : 1) * (WN_ROWS * 1024);
  asm volatile("" : "+v"(jwx), "+v"(jwy));
  Slot16 ta, tb, tc;                                      // the half-job's three raw slots
  unsigned tx[4], ty[4];
  auto job_read = [&](int i, int ks, int rawbuf) {
    const int kx = ks << 5;
    ta = *(const Slot16*)(smem + rawbuf + (ja[i] ^ kx));
    tb = *(const Slot16*)(smem + rawbuf + (jb[i] ^ kx));
    tc = *(const Slot16*)(smem + rawbuf + (jc[i] ^ kx));
  };
  float txl = 0.f, tyl = 0.f;                             // (bf16) low halves of the dword in flight
  auto job_lo = [&](int d) {
    if constexpr (DT != RD_F16) {
      const f32x2_t_ av = H16<DT>::unpk(ta[d]), bv = H16<DT>::unpk(tb[d]), cv = H16<DT>::unpk(tc[d]);
      txl = av[0] - cv[0];
      tyl = __builtin_fmaf(jsgn, cv[0], bv[0]);
    }
  };
  auto job_hi = [&](int d) {
    if constexpr (DT == RD_F16) {
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      const h2 av = __builtin_bit_cast(h2, (unsigned)ta[d]), bv = __builtin_bit_cast(h2, (unsigned)tb[d]), cv = __builtin_bit_cast(h2, (unsigned)tc[d]);
      const h2 sg = {(_Float16)jsgn, (_Float16)jsgn};
      tx[d] = __builtin_bit_cast(unsigned, av - cv);
      ty[d] = __builtin_bit_cast(unsigned, bv + sg * cv);
    } else {
      const f32x2_t_ av = H16<DT>::unpk(ta[d]), bv = H16<DT>::unpk(tb[d]), cv = H16<DT>::unpk(tc[d]);
      tx[d] = H16<DT>::pk(txl, av[1] - cv[1]);
      ty[d] = H16<DT>::pk(tyl, __builtin_fmaf(jsgn, cv[1], bv[1]));
    }
  };
  auto job_dword = [&](int d) { job_lo(d); job_hi(d); };
  auto job_write = [&](int i, int vhalf) {
    *(Slot16*)(smem + jwx + (vhalf * WN_VHALF + 2 * i * 1024)) = Slot16{tx[0], tx[1], tx[2], tx[3]};
    *(Slot16*)(smem + jwy + (vhalf * WN_VHALF + 2 * i * 1024)) = Slot16{ty[0], ty[1], ty[2], ty[3]};
  };

  // ---- MFMA operands ----------------------------------------------------------------------------------------------------------------
  int va = WN_V0 + lane * 16 + rp * 2048;                 // + ks * VHALF + (m * 6 + i) * 1024: V fragment of tile row 2rp + i
  int vb = WN_RING + lane * 16 + ch * 2048;               // + slot * SLAB + (dh * 4 + j) * 1024: U fragment of cout block 2ch + j
  asm volatile("" : "+v"(va), "+v"(vb));
  f32x16 acc[4][2][2];
  s16x8 fa[2][4], fb[2][6];
#define WN_RD(BUF, K, KS, MI, SLOT)                                                                              \
  {                                                                                                              \
    if ((K) < 4) fa[BUF][(K)] = *(const s16x8*)(smem + va + ((KS) * WN_VHALF + ((MI) * WN_ROWS + (K)) * 1024));  \
    else fb[BUF][(K) - 4] = *(const s16x8*)(smem + vb + ((SLOT) * WN_SLAB + ((((K) - 4) >> 1) * 4 + (((K) - 4) & 1)) * 1024)); \
    WN_FENCE();                                                                                                  \
  }
  // MFMA n of a step: dh = n >> 2, row = (n >> 1) & 1, cf = n & 1  (weights are operand A: the result's lanes are pixel pairs)
#define WN_MM(BUF, MI, N, ZERO)                                                                                   \
  {                                                                                                              \
    if (!(DBG & 8))                                                                                              \
      acc[MI][((N) >> 1) & 1][(N) & 1] = H16<DT>::mfma(fb[BUF][((N) >> 2) * 2 + ((N) & 1)], fa[BUF][(((N) >> 1) & 1) + ((N) >> 2)], \
                                                      (ZERO) ? f32x16{} : acc[MI][((N) >> 1) & 1][(N) & 1]);  \
    WN_FENCE();                                                                                                  \
  }
#define WN_SYNC(VMCNT, LGKM)                                       \
  {                                                               \
    asm volatile("" ::: "memory");                                \
    __builtin_amdgcn_s_waitcnt(WN_WAIT_IMM((DBG & 32) ? 63 : (VMCNT), LGKM)); \
    if (!(DBG & 2)) __builtin_amdgcn_s_barrier();                 \
    asm volatile("" ::: "memory");                                \
    WN_FENCE();                                                   \
  }

  // shift of this wave's two cout blocks as the A operand of a rank-1 MFMA (k = 0, 1: bf16 hi + lo), and its negative
  unsigned bzw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float t = a.shift ? a.shift[32 * (2 * ch + j) + conv_row_perm(m)] : 0.f;
    const bf16_t th = H16<DT>::from_f32(t);
    const bf16_t tl = H16<DT>::from_f32(t - H16<DT>::to_f32(th));
    bzw[j] = hi ? 0u : ((unsigned)th | ((unsigned)tl << 16));
  }

  // ---- prologue: raw halos of units 0 and 1, a full ring, V half 0 of unit 0 ------------------------------------------------------
  halo_begin();
#pragma unroll
  for (int j = 0; j < WN_PPW; ++j) halo_piece(0, j);
  halo_begin();
#pragma unroll
  for (int j = 0; j < WN_PPW; ++j) halo_piece(WN_RAWB, j);
#pragma unroll 1
  for (int s0 = 0; s0 < WN_R; ++s0) {
#pragma unroll
    for (int j = 0; j < 3; ++j) slab_piece(s0, j);
    slab_advance();
  }
  WN_SYNC(0, 0)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    job_read(i, 0, 0);
#pragma unroll
    for (int d = 0; d < 4; ++d) job_dword(d);
    job_write(i, 0);
  }
  WN_SYNC(0, 0)
  int rawcur = 0;                                         // byte offset of the raw buffer of the unit being computed
#pragma unroll
  for (int k = 0; k < 10; ++k) WN_RD(0, k, 0, 0, 0)

  // One step (ordinal O of its unit: ks = O >> 2, m = O & 3).  FIRST: first chunk of a tile -- the accumulators of m = 1, 2 start from
  // C = 0 (m = 0, 3 were started from +-shift).
#define WN_STEP(O, FIRST)                                                                                          \
  {                                                                                                                \
    constexpr int ks_ = (O) >> 2, mi_ = (O) & 3, on_ = ((O) + 1) & 7, nks_ = on_ >> 2, nmi_ = on_ & 3;             \
    constexpr int cur_ = (O) & 1, nxt_ = cur_ ^ 1;                                                                 \
    constexpr bool job_ = mi_ < 3 && !(DBG & 64);                                                                                 \
    constexpr bool z_ = (FIRST) && ks_ == 0 && (mi_ == 1 || mi_ == 2);                                             \
    constexpr int nraw_ = wn_nraw(O), rf_ = wn_raw_first(O);                                                       \
    const int jraw_ = ks_ == 0 ? rawcur : WN_RAWB - rawcur;   /* window ks 0 transforms raw(u) -> V half 1, ks 1 raw(u+1) -> V half 0 */ \
    WN_FENCE();                                                                                                    \
    if constexpr (job_) { job_read(mi_, ks_ ^ 1, jraw_); WN_FENCE(); }                                             \
    if constexpr ((O) == 4) halo_begin();                                                                          \
    _Pragma("unroll") for (int n = 0; n < 12; ++n) {                                                               \
      WN_MM(cur_, mi_, n, z_ && n < 4)                                                                             \
      if (n < 10) WN_RD(nxt_, n, nks_, nmi_, on_ & 3)                                                              \
      if constexpr (job_) {                                                                                        \
        if (n >= 1 && n <= 8) { if (n & 1) job_lo((n - 1) >> 1); else job_hi((n - 1) >> 1); WN_FENCE(); }          \
        if (n == 9) { job_write(mi_, ks_ ^ 1); WN_FENCE(); }                                                       \
      }                                                                                                            \
      if (!(DBG & 4)) {                                                                                            \
        if (nraw_ == 2 && n == 9) { halo_piece(rawcur, rf_); WN_FENCE(); }                                         \
        if (nraw_ >= 1 && n == 10) { halo_piece(rawcur, rf_ + nraw_ - 1); WN_FENCE(); }                            \
        if (n == 10) { slab_piece((O) & 3, 0); WN_FENCE(); }                                                       \
        if (n == 11) { slab_piece((O) & 3, 1); slab_piece((O) & 3, 2); WN_FENCE(); }                               \
      }                                                                                                            \
    }                                                                                                              \
    slab_advance();                                                                                                \
    WN_SYNC(wn_younger(O), (job_ && mi_ < 2) ? 2 : 0)                                                              \
  }
#define WN_CHUNK(FIRST) WN_STEP(0, FIRST) WN_STEP(1, FIRST) WN_STEP(2, FIRST) WN_STEP(3, FIRST) WN_STEP(4, FIRST) WN_STEP(5, FIRST) WN_STEP(6, FIRST) WN_STEP(7, FIRST)

  for (int k = 0; k < ntl; ++k) {
    {   // accumulators of m = 0 / 3 start from + / - shift (y0 = M0 + M1 + M2, y1 = M1 - M2 - M3)
      unsigned z0 = 0u;
      asm volatile("" : "+v"(z0));
      const unsigned one2 = hi ? z0 : H16<DT>::ONE * 0x10001u;
      unsigned ob[4] = {one2, z0, z0, z0};
      s16x8 ones;
      memcpy(&ones, ob, 16);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        unsigned ab[4] = {bzw[n & 1], z0, z0, z0}, nb[4] = {bzw[n & 1] ^ (hi ? 0u : 0x80008000u), z0, z0, z0};
        s16x8 bz, nz;
        memcpy(&bz, ab, 16);
        memcpy(&nz, nb, 16);
        acc[0][n >> 1][n & 1] = H16<DT>::mfma(bz, ones, f32x16{});
        acc[3][n >> 1][n & 1] = H16<DT>::mfma(nz, ones, f32x16{});
      }
      WN_FENCE();
    }
    WN_CHUNK(true)
    rawcur = WN_RAWB - rawcur;
#pragma unroll 1
    for (int c = 1; c < a.nchunk; ++c) {
      WN_CHUNK(false)
      rawcur = WN_RAWB - rawcur;
    }

    // ---- epilogue: output transform, residual, ReLU, one rounding; transpose through the free V half 1 (6 KB per wave) so that every
    // store instruction writes whole 128-byte halves of pixel rows.  Lane (p, hi) of acc[mi][r][j] holds channels 64 ch + 32 j + 16 hi + q.
    int ct, rb, b;
    decode(wg + k * G, ct, rb, b);
    int em = m, ehi = hi, el = lane;
    asm volatile("" : "+v"(em), "+v"(ehi), "+v"(el));
    const bool relu_pre = (a.flags & RD_RELU_PRE) != 0, do_add = (a.flags & RD_ADD) != 0, relu_post = (a.flags & RD_RELU_POST) != 0;
    const bool relu_f32 = relu_pre && do_add, relu_i16 = relu_post || (relu_pre && !do_add);
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    unsigned char* scr = smem + WN_V0 + WN_VHALF + wave * 6144;
    const int w0 = ct * WN_TW;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int oh = rb * WN_TH + 2 * rp + r;
      bf16_t* __restrict__ yrow = a.y + (size_t)b * a.y_bs + (size_t)oh * a.W * a.y_cs + a.y_co + 64 * ch;
      const bf16_t* __restrict__ rrow = a.res + (size_t)b * a.r_bs + a.r_co + 64 * ch + 16 * ehi;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        Slot16 rv[2][2];
        if (do_add) {
          const int ow = w0 + 2 * em + e;
          const bool live = ow < a.W && oh < a.H;
          const bf16_t* rq = rrow + (live ? ((size_t)oh * a.W + ow) * a.r_cs : 0);
#pragma unroll
          for (int j = 0; j < 2; ++j) { rv[j][0] = *(const Slot16*)(rq + 32 * j); rv[j][1] = *(const Slot16*)(rq + 32 * j + 8); }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          WN_FENCE();
          unsigned pk[8];
#pragma unroll
          for (int q2 = 0; q2 < 8; ++q2) {
            float v0, v1;
            if (e == 0) {
              v0 = (acc[0][r][j][2 * q2] + acc[1][r][j][2 * q2]) + acc[2][r][j][2 * q2];
              v1 = (acc[0][r][j][2 * q2 + 1] + acc[1][r][j][2 * q2 + 1]) + acc[2][r][j][2 * q2 + 1];
            } else {
              v0 = (acc[1][r][j][2 * q2] - acc[2][r][j][2 * q2]) - acc[3][r][j][2 * q2];
              v1 = (acc[1][r][j][2 * q2 + 1] - acc[2][r][j][2 * q2 + 1]) - acc[3][r][j][2 * q2 + 1];
            }
            if (relu_f32) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
            if (do_add) {
              const f32x2_t_ rr = H16<DT>::unpk(rv[j][q2 >> 2][q2 & 3]);
              v0 += rr[0]; v1 += rr[1];
            }
            unsigned p2 = H16<DT>::pk(v0, v1);
            if (relu_i16) p2 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p2), (s16x2){0, 0}));
            pk[q2] = p2;
          }
#pragma unroll
          for (int u = 0; u < 2; ++u)
            *(Slot16*)(scr + em * 128 + (((4 * j + 2 * ehi + u) ^ (em & 7)) << 4)) = Slot16{pk[4 * u], pk[4 * u + 1], pk[4 * u + 2], pk[4 * u + 3]};
        }
        WN_FENCE();
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int pr = it * 8 + (el >> 3), sl = el & 7;
          const Slot16 v = *(const Slot16*)(scr + pr * 128 + ((sl ^ (pr & 7)) << 4));
          const int ow = w0 + 2 * pr + e;
          if (ow < a.W && oh < a.H && !(DBG & 1)) __builtin_nontemporal_store(v, (Slot16*)(yrow + (size_t)ow * a.y_cs + sl * 8));
        }
        __builtin_amdgcn_wave_barrier();
        WN_FENCE();
      }
    }
    // the next tile's first transform step writes V half 1: every wave must be done with its scratch
    WN_SYNC(63, 0)
  }
  __builtin_amdgcn_s_waitcnt(RD_VMCNT_IMM(0));   // the dummy tail fetches target this workgroup's LDS
#undef WN_CHUNK
#undef WN_STEP
#undef WN_SYNC
#undef WN_MM
#undef WN_RD
#undef WN_FENCE
}

inline int wino_num_cus() {
#ifdef RD_BUILD_NUM_CUS
  return RD_BUILD_NUM_CUS;
#else
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  return v;
#endif
}

template <int DT, int DBG>
inline int launch_wino_dbg(const WinoArgs& a, hipStream_t st) {
  auto k = wino3x3_stream_kernel<DT, DBG>;
  static std::atomic<unsigned long long> seen{0};
  once_per_device(seen, [&] { allow_big_lds(k); });
  const int grid = std::min(a.ntiles, wino_num_cus());
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), WN_LDS, st, a);
  return check_launch("wino3x3_stream_kernel");
}
template <int DT>
inline int launch_wino_dt(const WinoArgs& a, hipStream_t st) {
#ifdef WN_ABLATIONS   // WINO_DBG=n (harness only): 4 no DMA after the prologue, 8 no MFMAs, 64 no transform jobs, 2 no barriers, 1 no stores
  static const int dbg = getenv("WINO_DBG") ? atoi(getenv("WINO_DBG")) : 0;
  if (DT == RD_BF16) {
    switch (dbg) {
      case 4: return launch_wino_dbg<RD_BF16, 4>(a, st);
      case 8: return launch_wino_dbg<RD_BF16, 8>(a, st);
      case 64: return launch_wino_dbg<RD_BF16, 64>(a, st);
      case 68: return launch_wino_dbg<RD_BF16, 68>(a, st);
      case 70: return launch_wino_dbg<RD_BF16, 70>(a, st);
      case 12: return launch_wino_dbg<RD_BF16, 12>(a, st);
      case 76: return launch_wino_dbg<RD_BF16, 76>(a, st);
      default: break;
    }
  }
#endif
  return launch_wino_dbg<DT, 0>(a, st);
}

// x (B, H, W, x_cs) 16-bit, cin % 32 == 0, cout = 128; w = pack_wino_frag image; shift (128) or null; res / y like rd_conv3x3_bn_act_ex
inline int launch_wino(const void* x, int x_cs, int x_co, const void* w, const float* shift, const void* res, int r_cs, int r_co, void* y,
                       int y_cs, int y_co, int B, int H, int W, int cin, int flags, int dt, hipStream_t st) {
  RD_REQUIRE(is_h16(dt), RD_EINVAL, "wino: dtype %d", dt);
  RD_REQUIRE(cin > 0 && cin % 32 == 0 && x_co % 8 == 0 && x_cs % 8 == 0 && x_co + cin <= x_cs, RD_ESHAPE, "wino: cin %d must be a multiple of 32 inside the pixel pitch %d", cin, x_cs);
  RD_REQUIRE(y_cs % 64 == 0 && y_co % 64 == 0 && y_co + 128 <= y_cs, RD_ESHAPE, "wino: output channel offset / pitch must be multiples of 64");
  RD_REQUIRE(!(flags & RD_ADD) || (res && r_cs % 8 == 0 && r_co % 8 == 0 && r_co + 128 <= r_cs), RD_EINVAL, "wino: RD_ADD needs a residual tensor");
  WinoArgs a;
  memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.x_cs = x_cs; a.x_co = x_co; a.x_bs = (long)H * W * x_cs;
  a.w = (const unsigned char*)w; a.shift = shift;
  a.res = (const bf16_t*)res; a.r_cs = r_cs; a.r_co = r_co; a.r_bs = (long)H * W * r_cs;
  a.y = (bf16_t*)y; a.y_cs = y_cs; a.y_co = y_co; a.y_bs = (long)H * W * y_cs;
  a.zero16 = (const unsigned char*)w + wino_packed_body_bytes(cin);
  a.H = H; a.W = W; a.B = B; a.nchunk = cin / 32; a.flags = flags & ~RD_SCALE_FOLDED;
  a.ncol = (W + WN_TW - 1) / WN_TW; a.nrow = (H + WN_TH - 1) / WN_TH; a.ntiles = a.ncol * a.nrow * B;
  ProfScope ps(RD_PROF_CONV3, st);
  if (dt == RD_F16) return launch_wino_dt<RD_F16>(a, st);
  return launch_wino_dt<RD_BF16>(a, st);
}

}  // namespace rd
