// EXPERIMENT (round 5, tools/micro only -- not part of librangedet_hip): the fused BasicBlock kernel of csrc/k_block.h on 16 x 32 tiles with
// ONE 4-wave workgroup per CU.  What the larger tile buys: conv1 on 18 x 34 = 612 positions enumerated DENSELY (20 fragments of 32, five per
// wave -- wave w: positions 160 w .. 160 w + 159) against 16 conv2 fragments = 1.125 x the block's MFMAs instead of 1.25 x; x with a
// 20 x 36 halo (1.41 x instead of 1.69 x); half the weight-slab DMA per pixel; 7 LDS reads per 10 MFMAs / 6 per 8 instead of 5 per 6 / 4 per 4.
// What it costs: 2 x 48 KB of x / t + an 8-deep ring = 128 KB of LDS, so one workgroup per CU (one wave per SIMD, 512 registers): nothing
// hides its barrier waits, DMA-issue stalls, the t write and the epilogue.  Identity shortcut, 64 -> 64 -> 64 only.
// x layout: 36-pixel pitch, 16-byte slots swizzled by COLUMN ((c >> 2) & 3) -- a dense fragment crosses rows (34 of 36 columns), and the tap
// row then shifts every lane by one pitch without changing its swizzle term.
#pragma once
#include "../../rangedet_amd/csrc/k_block.h"

namespace rd {

constexpr int BT_R = 8;
constexpr int BT_XPIECES = 12;                  // 1-KB DMA pieces per wave and x chunk: 48 pieces = 768 pixels >= 20 x 36 = 720
constexpr int BT_BUF = 4 * BT_XPIECES * 1024;   // 49 152
constexpr size_t BT_LDS = 2 * BT_BUF + BT_R * BK_SLAB;   // 131 072
constexpr int bt_pieces_u(int s) { return s <= 1 ? 3 : s <= 4 ? 2 : 0; }
constexpr int bt_first_u(int s) { int n = 0; for (int t = 0; t < s; ++t) n += bt_pieces_u(t); return n; }
constexpr int bt_pieces(int g) { const int u = g / 9; return (u == 0 || u == 3) ? bt_pieces_u(g % 9) : 0; }
constexpr int bt_younger(int g) {
  int n = BT_R - 3;
  for (int d = 1; d <= BT_R - 2; ++d) n += bt_pieces(((g - d) % 36 + 36) % 36);
  const int u = g / 9, s = g % 9, cap = 9 - 3 - 4;
  return ((u == 0 || u == 3) && s == 7 && n > cap) ? cap : n;
}

template <int DT>
__global__ __launch_bounds__(256, 1) void block64_tall_kernel(BlockArgs a) {
  HIP_DYNAMIC_SHARED(unsigned char, smem);
  constexpr int R = BT_R, SLAB = BK_SLAB, RING = 2 * BT_BUF, NCT = 2;
  constexpr int ROWB2 = BK_TP * 64, ROWB1 = BK_XP * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, hi = lane >> 5;
  unsigned bz1[NCT], bz2[NCT];
#pragma unroll
  for (int j = 0; j < NCT; ++j) {
    const float t1 = a.shift1[32 * j + conv_row_perm(m)], t2 = a.shift2[32 * j + conv_row_perm(m)];
    const bf16_t h1 = H16<DT>::from_f32(t1), h2 = H16<DT>::from_f32(t2);
    const bf16_t l1 = H16<DT>::from_f32(t1 - H16<DT>::to_f32(h1)), l2 = H16<DT>::from_f32(t2 - H16<DT>::to_f32(h2));
    bz1[j] = hi ? 0u : ((unsigned)h1 | ((unsigned)l1 << 16));
    bz2[j] = hi ? 0u : ((unsigned)h2 | ((unsigned)l2 << 16));
  }
  const int G = gridDim.x, wg = blockIdx.x;
  const int ntl = (a.ntiles - wg + G - 1) / G;
  const int tiles_img = a.ncol * a.nrow;
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem);
#endif
  auto dma_s = [&](const unsigned char* sbase, unsigned voff, int lds_off) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds0 + (unsigned)lds_off)) : "memory");
#else
    __builtin_amdgcn_global_load_lds(sbase + voff, smem + lds_off, 16, 0, 0);
#endif
  };
  auto dma_v = [&](const void* vptr, int lds_off) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(vptr), "s"(__builtin_amdgcn_readfirstlane(lds0 + (unsigned)lds_off)) : "memory");
#else
    __builtin_amdgcn_global_load_lds(vptr, smem + lds_off, 16, 0, 0);
#endif
  };
  // tile cursor: plain order (column tile fastest)
  const int g_ct = G % a.ncol, g_rb = (G / a.ncol) % a.nrow, g_b = G / tiles_img;
  auto tile_advance = [&](int& ct, int& rb, int& b) __attribute__((always_inline)) {
    ct += g_ct;
    const int c1 = ct >= a.ncol ? 1 : 0;
    ct -= c1 ? a.ncol : 0;
    rb += g_rb + c1;
    const int c2 = rb >= a.nrow ? 1 : 0;
    rb -= c2 ? a.nrow : 0;
    b += g_b + c2;
  };
  int f_ct = wg % a.ncol, f_rb = (wg / a.ncol) % a.nrow, f_b = wg / tiles_img;
  int c_ct = f_ct, c_rb = f_rb, c_b = f_b;
  int hk = 0, hc = 0, hh0 = 0, hw0 = 0;
  const unsigned char* htile = nullptr;
  const unsigned char* hbase = nullptr;
  bool hnew = true;
  auto halo_begin = [&]() {
    if (hnew) {
      hh0 = f_rb * 16 - 2; hw0 = f_ct * 32 - 2;
      htile = (const unsigned char*)(a.x + (size_t)f_b * a.x_bs + a.x_co) + ((long)hh0 * a.W + hw0) * (long)a.x_cs * 2;
      hnew = false;
    }
    hbase = htile + hc * 64;
    if (hc == 0) hc = 1;
    else if (hk + 1 < ntl) { ++hk; hc = 0; tile_advance(f_ct, f_rb, f_b); hnew = true; }
  };
  auto halo_piece = [&](int buf, int j) {
    const int q = wave * BT_XPIECES + j;
    int ol = lane;
    asm volatile("" : "+v"(ol));
    const int pp = 16 * q + (ol >> 2), r = (pp * 1821) >> 16, cc = pp - BK_XP * r;      // pp / 36 exactly for pp < 768
    const int hs = (ol & 3) ^ ((cc >> 2) & 3);                      // logical slot: physical slot ^ column swizzle
    const bool ok = pp < 20 * BK_XP && (unsigned)(hh0 + r) < (unsigned)a.H && (unsigned)(hw0 + cc) < (unsigned)a.W;
    const unsigned char* src = hbase + (long)((r * a.W + cc) * a.x_cs * 2 + hs * 16);
    dma_v(ok ? (const void*)src : (const void*)a.zero16, buf + q * 1024);
  };
  int fslot = 0, fslab = 0;
  auto slab_piece = [&]() { dma_s(a.w + (size_t)fslab * SLAB + wave * 1024, lane * 16, RING + fslot * SLAB + wave * 1024); };
  auto slab_advance = [&]() {
    fslot = fslot + 1 == R ? 0 : fslot + 1;
    fslab = fslab + 1 == 36 ? 0 : fslab + 1;
  };
  // conv1 (dense): fragment i of the wave, lane m: position n = 160 w + 32 i + m of the 18 x 34 t grid = (r', c'); tap (dh, dw) reads x
  //   (r' + dh, c' + dw): byte = 64 (36 (r' + dh) + c' + dw) + (((2 ks + hi) ^ ((c' + dw) >> 2)) & 3) * 16 -> a1[i][dw] + dh * ROWB1, ^ 32 for ks 1
  // conv2 (rows): output row 4 w + i, column m: t pixel (4 w + i + dh, m + dw) -> a2[dw] + (i + dh) * ROWB2
  int a1[5][3], a2[3];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int n = 160 * wave + 32 * i + m, rr = (n * 1928) >> 16, cc = n - BK_TP * rr;      // n / 34 exactly for n < 640
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) a1[i][dw] = (BK_XP * rr + cc + dw) * 64 + (((hi ^ ((cc + dw) >> 2)) & 3) << 4);
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int c = d + m;
    a2[d] = c * 64 + (((hi ^ (c >> 2)) & 3) << 4) + wave * 4 * ROWB2;
  }
  const int boff = RING + lane * 16;
  f32x16 acc[5][NCT];
  s16x8 fa[2][5], fb[2][NCT];
#define BK_FENCE() __builtin_amdgcn_sched_barrier(0)
  // read k of a k-step.  KIND 1 (conv1, 7 reads): fa[0], fb[0], fb[1], fa[1..4] at a1[i][DW] + A (A = dh * ROWB1 + buffer), ^ X;
  // KIND 2 (conv2, 6 reads): fa[0], fb[0], fb[1], fa[1..3] at A + i * ROWB2 (A = full address incl. ^ 32)
#define BT_RD(KIND, BUF, K, DW, A, X, BADDR, KS)                                                            \
  {                                                                                                        \
    if ((K) >= 1 && (K) <= NCT) fb[BUF][(K) - 1] = *(const s16x8*)(smem + (BADDR) + ((KS) * NCT + (K) - 1) * 1024); \
    else if ((KIND) == 1) fa[BUF][(K) == 0 ? 0 : ((K) - NCT) % 5] = *(const s16x8*)(smem + ((a1[(K) == 0 ? 0 : ((K) - NCT) % 5][DW] + (A)) ^ (X))); \
    else fa[BUF][(K) == 0 ? 0 : ((K) - NCT) % 5] = *(const s16x8*)(smem + (A) + ((K) == 0 ? 0 : (K) - NCT) * ROWB2); \
    BK_FENCE();                                                                                            \
  }
#define BK_MM(BUF, N)                                                                                      \
  {                                                                                                        \
    acc[(N) / NCT][(N) % NCT] = H16<DT>::mfma(fb[BUF][(N) % NCT], fa[BUF][(N) / NCT], acc[(N) / NCT][(N) % NCT]); \
    BK_FENCE();                                                                                            \
  }
#define BK_SYNC(VMCNT, LGKM)                                      \
  {                                                               \
    asm volatile("" ::: "memory");                                \
    __builtin_amdgcn_s_waitcnt(C3_WAIT_IMM(VMCNT, LGKM));         \
    __builtin_amdgcn_s_barrier();                                 \
    asm volatile("" ::: "memory");                                \
    BK_FENCE();                                                   \
  }
  // One step G_ of the tile.  CK_ = kind of this step (1: conv1, 5 fragments; 2: conv2, 4), its k-step-1 reads from (CDW_, CA_, CX_);
  // NK_ = kind of the next step's k-step-0 reads (0: none -- the pipeline is cut), from (NDW_, NA_).  HB_ = buffer of the halo pieces.
#define BT_STEP(G_, HB_, CK_, CDW_, CA_, CX_, NK_, NDW_, NA_)                                                      \
  {                                                                                                                \
    constexpr int F_ = (CK_) == 1 ? 5 : 4, NM_ = F_ * NCT, NR_ = F_ + NCT, NRN_ = (NK_) == 0 ? 0 : ((NK_) == 1 ? 5 : 4) + NCT; \
    constexpr int NH_ = bt_pieces(G_), NP_ = 1 + NH_, HF_ = bt_first_u((G_) % 9), YG_ = bt_younger(G_);            \
    const int bcur_ = boff + rslot * SLAB;                                                                         \
    const int rnext_ = rslot + 1 == R ? 0 : rslot + 1;                                                             \
    const int bnext_ = boff + rnext_ * SLAB;                                                                       \
    const int ca_ = (CA_), na_ = (NA_);                                                                            \
    BK_FENCE();                                                                                                    \
    _Pragma("unroll") for (int n = 0; n < NM_; ++n) {                                                              \
      BK_MM(0, n)                                                                                                  \
      if (n < NR_) BT_RD(CK_, 1, n, CDW_, ca_, CX_, bcur_, 1)                                                      \
    }                                                                                                              \
    _Pragma("unroll") for (int n = 0; n < NM_ / 2; ++n) {                                                          \
      BK_MM(1, n)                                                                                                  \
      if (n < NRN_) BT_RD(NK_, 0, n, NDW_, na_, 0, bnext_, 0)                                                      \
    }                                                                                                              \
    { _Pragma("unroll") for (int n = NM_ / 2; n < NRN_; ++n) BT_RD(NK_, 0, n, NDW_, na_, 0, bnext_, 0) }           \
    BK_SYNC(YG_, NRN_)                                                                                             \
    if (NH_ > 0 && (G_) % 9 == 0) halo_begin();                                                                    \
    _Pragma("unroll") for (int n = NM_ / 2; n < NM_; ++n) {                                                        \
      BK_MM(1, n)                                                                                                  \
      _Pragma("unroll") for (int p = 0; p < NP_; ++p)                                                              \
        if (p * (NM_ / 2) / NP_ == n - NM_ / 2) {                                                                  \
          if (p == 0) slab_piece(); else halo_piece(HB_, HF_ + p - 1);                                             \
          BK_FENCE();                                                                                              \
        }                                                                                                          \
    }                                                                                                              \
    slab_advance();                                                                                                \
    rslot = rnext_;                                                                                                \
  }
  // conv1 step (U_, S_): tap S_ of x chunk U_ in buffer U_; next: tap S_ + 1, or tap 0 of buffer 1 after U0, or nothing after U1
#define BT_C1(U_, S_)                                                                                              \
  BT_STEP(9 * (U_) + (S_), BT_BUF, 1, (S_) % 3, ((S_) / 3) * ROWB1 + (U_) * BT_BUF, 32,                            \
          (((U_) == 1 && (S_) == 8) ? 0 : 1), (((S_) + 1) % 9) % 3, ((S_) == 8 ? BT_BUF : ((((S_) + 1) % 9) / 3) * ROWB1 + (U_) * BT_BUF))
  // conv2 step (C_, S_): tap S_ of t chunk C_ in buffer C_; next: tap S_ + 1, or tap 0 of buffer 1 after U2, or the NEXT tile's conv1 tap 0
#define BT_C2(C_, S_)                                                                                              \
  BT_STEP(18 + 9 * (C_) + (S_), 0, 2, 0, (a2[(S_) % 3] + ((S_) / 3) * ROWB2 + (C_) * BT_BUF) ^ 32, 0,              \
          (((C_) == 1 && (S_) == 8) ? 1 : 2), 0,                                                                   \
          ((S_) == 8 ? ((C_) == 1 ? 0 : a2[0] + BT_BUF) : a2[((S_) + 1) % 3] + ((((S_) + 1) % 9) / 3) * ROWB2 + (C_) * BT_BUF))

  halo_begin();
#pragma unroll
  for (int j = 0; j < BT_XPIECES; ++j) halo_piece(0, j);
#pragma unroll 1
  for (int s0 = 0; s0 < R; ++s0) { slab_piece(); slab_advance(); }
  BK_SYNC(0, 0)
  int rslot = 0;
#pragma unroll
  for (int kk = 0; kk < 5 + NCT; ++kk) BT_RD(1, 0, kk, 0, 0, 0, boff + rslot * SLAB, 0)

  for (int k = 0; k < ntl; ++k) {
    unsigned z0 = 0u;
    asm volatile("" : "+v"(z0));
    const unsigned one2 = hi ? z0 : H16<DT>::ONE * 0x10001u;
    s16x8 ones;
    {
      unsigned ob[4] = {one2, z0, z0, z0};
      memcpy(&ones, ob, 16);
    }
#pragma unroll
    for (int n = 0; n < 5 * NCT; ++n) {
      unsigned ab[4] = {bz1[n % NCT], z0, z0, z0};
      s16x8 bz;
      memcpy(&bz, ab, 16);
      acc[n / NCT][n % NCT] = H16<DT>::mfma(bz, ones, f32x16{});
    }
    BK_FENCE();
    BT_C1(0, 0) BT_C1(0, 1) BT_C1(0, 2) BT_C1(0, 3) BT_C1(0, 4) BT_C1(0, 5) BT_C1(0, 6) BT_C1(0, 7) BT_C1(0, 8)
    BT_C1(1, 0) BT_C1(1, 1) BT_C1(1, 2) BT_C1(1, 3) BT_C1(1, 4) BT_C1(1, 5) BT_C1(1, 6) BT_C1(1, 7) BT_C1(1, 8)

    // t = relu(conv1 + shift1), rounded, into buf0 / buf1: position n of the dense 18 x 34 grid IS its pixel index in the 34-pitch image
    const int ct = c_ct, rb = c_rb, b = c_b;
    tile_advance(c_ct, c_rb, c_b);
    {
      int em = m, ehi = hi;
      asm volatile("" : "+v"(em), "+v"(ehi));
      typedef short s16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int n = 160 * wave + 32 * i + em, rr = (n * 1928) >> 16, cc = n - BK_TP * rr;
        const bool keep = n < 18 * BK_TP;
        const bool inside = (unsigned)(rb * 16 - 1 + rr) < (unsigned)a.H && (unsigned)(ct * 32 - 1 + cc) < (unsigned)a.W;
        const int doff = n * 64 + ((((2 * ehi) ^ (cc >> 2)) & 3) << 4);
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
          BK_FENCE();
          unsigned pk[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            unsigned p2 = H16<DT>::pk(acc[i][j][2 * q], acc[i][j][2 * q + 1]);
            p2 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p2), (s16x2){0, 0}));
            pk[q] = inside ? p2 : 0u;
          }
          if (keep) {
            *(Slot16*)(smem + j * BT_BUF + doff) = Slot16{pk[0], pk[1], pk[2], pk[3]};
            *(Slot16*)(smem + j * BT_BUF + (doff ^ 16)) = Slot16{pk[4], pk[5], pk[6], pk[7]};
          }
        }
      }
    }
#pragma unroll
    for (int n = 0; n < 4 * NCT; ++n) {
      unsigned ab[4] = {bz2[n % NCT], z0, z0, z0};
      s16x8 bz;
      memcpy(&bz, ab, 16);
      acc[n / NCT][n % NCT] = H16<DT>::mfma(bz, ones, f32x16{});
    }
    BK_FENCE();
    BK_SYNC(63, 0)
#pragma unroll
    for (int kk = 0; kk < 4 + NCT; ++kk) BT_RD(2, 0, kk, 0, a2[0], 0, boff + rslot * SLAB, 0)
    BT_C2(0, 0) BT_C2(0, 1) BT_C2(0, 2) BT_C2(0, 3) BT_C2(0, 4) BT_C2(0, 5) BT_C2(0, 6) BT_C2(0, 7) BT_C2(0, 8)
    BT_C2(1, 0) BT_C2(1, 1) BT_C2(1, 2) BT_C2(1, 3) BT_C2(1, 4) BT_C2(1, 5) BT_C2(1, 6) BT_C2(1, 7) BT_C2(1, 8)

    {
      const int oh0 = rb * 16 + 4 * wave;
      int em = m, ehi = hi, el = lane;
      asm volatile("" : "+v"(em), "+v"(ehi), "+v"(el));
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      unsigned char* scr = smem + BT_BUF + wave * (BT_BUF / 4);
      bf16_t* __restrict__ yrow0 = a.y + (size_t)b * a.y_bs + (size_t)oh0 * a.W * a.y_cs + a.y_co;
      const bf16_t* __restrict__ rimg0 = a.x + (size_t)b * a.x_bs + a.x_co;
      Slot16 rv[2][NCT][2];
      auto res_load = [&](int i, Slot16 (&dst)[NCT][2]) {
        const int ow = ct * 32 + em, oh = oh0 + i;
        const bool live = ow < a.W && oh < a.H;
        const bf16_t* rp = rimg0 + (live ? ((size_t)oh * a.W + (size_t)ow) * a.x_cs : 0) + 16 * ehi;
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
          dst[j][0] = *(const Slot16*)(rp + j * 32);
          dst[j][1] = *(const Slot16*)(rp + j * 32 + 8);
        }
      };
      res_load(0, rv[0]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i + 1 < 4) res_load(i + 1, rv[(i + 1) & 1]);
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
          BK_FENCE();
          const int cb = j * 32 + 16 * ehi;
          unsigned pk[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            f32x2 v = {acc[i][j][2 * q], acc[i][j][2 * q + 1]};
            v += H16<DT>::unpk(rv[i & 1][j][q >> 2][q & 3]);
            unsigned p2 = H16<DT>::pk(v[0], v[1]);
            pk[q] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p2), (s16x2){0, 0}));
          }
#pragma unroll
          for (int u = 0; u < 2; ++u)
            *(Slot16*)(scr + em * 128 + ((((cb >> 3) + u) ^ (em & 7)) << 4)) = Slot16{pk[4 * u], pk[4 * u + 1], pk[4 * u + 2], pk[4 * u + 3]};
        }
        BK_FENCE();
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int pr = it * 8 + el / 8, sl = el % 8;
          const Slot16 v = *(const Slot16*)(scr + pr * 128 + ((sl ^ (pr & 7)) << 4));
          const int ows = ct * 32 + pr;
          if (ows < a.W && oh0 + i < a.H)
            __builtin_nontemporal_store(v, (Slot16*)(yrow0 + (size_t)i * a.W * a.y_cs + (size_t)ows * a.y_cs + sl * 8));
        }
        __builtin_amdgcn_wave_barrier();
        BK_FENCE();
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(RD_VMCNT_IMM(0));
#undef BT_C2
#undef BT_C1
#undef BT_STEP
#undef BK_SYNC
#undef BK_MM
#undef BT_RD
#undef BK_FENCE
}

inline int launch_block64_tall(const void* x, int x_cs, int x_co, const void* w, const float* shift1, const float* shift2, void* y, int y_cs,
                               int y_co, int B, int H, int W, int dt, hipStream_t st) {
  RD_REQUIRE(dt == RD_BF16, RD_EINVAL, "block64_tall: bf16 only (experiment)");
  BlockArgs a;
  memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.x_cs = x_cs; a.x_co = x_co; a.x_bs = (long)H * W * x_cs;
  a.w = (const unsigned char*)w; a.shift1 = shift1; a.shift2 = shift2;
  a.y = (bf16_t*)y; a.y_cs = y_cs; a.y_co = y_co; a.y_bs = (long)H * W * y_cs;
  a.zero16 = (const unsigned char*)w + block64_body_bytes(64);
  a.H = H; a.W = W; a.B = B;
  a.ncol = (W + 31) / 32; a.nrow = (H + 15) / 16; a.ntiles = a.ncol * a.nrow * B;
  const int grid = std::min(a.ntiles, conv_num_cus());
  allow_big_lds(block64_tall_kernel<RD_BF16>);
  hipLaunchKernelGGL((block64_tall_kernel<RD_BF16>), dim3(grid), dim3(256), BT_LDS, st, a);
  return check_launch("block64_tall_kernel");
}

}  // namespace rd
