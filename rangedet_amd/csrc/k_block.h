// A whole 64-channel BasicBlock at stride 1 (dla_backbone.py:18-56) as ONE persistent launch:
//     t = relu(BN1(conv1_3x3(x)))        y = relu(BN2(conv2_3x3(t)) + x)                     (units 2.. of a stage)
//                                        y = relu(BN2(conv2_3x3(t)) + BNs(conv_1x1(x)))      (SC: unit 1, projection shortcut :44-51)
// (FIRST: the network's first block, x with <= 16 channels: conv1 as five two-tap steps on one 16-channel k-slot instead of U0 / U1 below.)
// The unfused pair (k_conv3.h, twice) moves 5.5 tile-sized passes through HBM per block (x with its halo, t written, t read back
// with its halo, the residual x again, y); here t never leaves the CU: conv1 is evaluated on the 10 x 34 pixels conv2 needs for an
// 8 x 32 output tile and its rounded result is written straight into LDS in the halo-image layout conv2 reads.  Price: conv1 runs
// on 12 instead of 8 pixel fragments per tile (+25 % of the block's MFMAs), and x is fetched with a 12 x 36 halo.
//
// Same machinery as conv3x3_stream_kernel<2, .., FPW 2, FC 1, WD> (8 x 32 tiles, two 4-wave workgroups per CU, LDS-DMA with counted
// waits, one workgroup barrier per step placed mid-block, fragment reads interleaved 1:1 with the MFMAs); what differs:
//   * a tile is FOUR units of 9 steps: U0 = conv1 chunk 0, U1 = conv1 chunk 1, U2 = conv2 chunk 0, U3 = conv2 chunk 1 (chunk = 32
//     input channels); the weight-slab stream is the 36 slabs [conv1 image | conv2 image] per tile through one 6-deep ring;
//   * two 28-KB LDS buffers, used alternately by the units: x chunk 0 -> buf0 (fetched during the PREVIOUS tile's U3), x chunk 1 ->
//     buf1 (fetched during U0); after U1 every wave writes its part of t = both 32-channel chunks of the intermediate into buf0 / buf1
//     (x is dead by then: 2 x 28 KB hold x and then t, so LDS stays at 80 KB and two workgroups share a CU); U2 / U3 read them like
//     any halo image; the epilogue's transpose scratch is buf1 again.  Units U1 and U2 issue no halo DMA.
//   * conv1's pixel fragments are FLAT: the x image has a row pitch of 36 pixels (12 rows x 36 columns) and fragment f covers the 32
//     consecutive flat positions 32 f .. 32 f + 31 of the 10 x 36 grid of t (tap (dh, dw) = +36 dh + dw), so 12 fragments cover all
//     340 needed positions (wave w: fragments 3 w .. 3 w + 2; positions in columns 34, 35 and past row 9 are computed and dropped).
//     The 16-byte slots of a pixel are XOR-swizzled with (flat position >> 2) & 3 -- conflict-free for any alignment of a fragment;
//   * t is stored with the 34-pixel pitch and the by-column swizzle of the wide tile (k_conv3.h C3Cfg<.., WD>), pixels outside the
//     image as ZERO (conv2's zero padding -- not conv1 evaluated on padding), so conv2's addressing is the production kernel's;
//   * the software pipeline is cut once per tile, after U1: t does not exist before every wave has written its part, so U1's last
//     step pre-reads nothing and U2's first fragments are read after the intermediate write (the other workgroup of the CU runs
//     under this bubble).  U3's last step pre-reads the NEXT tile's first conv1 fragments (3 instead of 2: the shape changes there).
// Results are BIT-IDENTICAL to the two launches of conv3x3_stream_kernel: every accumulator sees the same MFMA sequence (shift,
// then chunk 0 taps 0..8, chunk 1 taps 0..8, two k-steps each) on the same operands, and t is rounded once, like the stored tensor.
#pragma once
#include "k_conv3.h"

namespace rd {

struct BlockArgs {
  const bf16_t* x; int x_cs, x_co; long x_bs;      // block input = residual (64 channels)
  const unsigned char* w;                          // [conv1: pack_taps_frag(9, 64, 64) (FIRST: pack_body_frag(3, cin <= 16, 64)) | conv2: pack_taps_frag | RD_CONV_TAIL zeros]
  const float* shift1; const float* shift2;        // BatchNorm shifts (scales folded into the weights; SC: shift2 = conv2's + the shortcut's)
  const unsigned char* scw;                        // SC: packed 1x1 shortcut weights, pack_sc_frag(64 -> 64) = [4 k-steps][2][64 lanes][8]
  bf16_t* y; int y_cs, y_co; long y_bs;
  const unsigned char* zero16;
  int H, W, B, ncol, nrow, ntiles, xcd;
  int nslots1;                                     // FIRST: valid 16-byte channel slots of x (1 or 2); the others are read from the zero page
};

constexpr int BK_R = 6;                    // ring depth (slabs of 4 KB)
constexpr int BK_XP = 36;                  // row pitch of the x image (pixels)
constexpr int BK_XPIECES = 7;              // 1-KB DMA pieces per wave and x chunk: 28 pieces = 448 pixels >= 12 x 36 = 432
constexpr int BK_BUF = 4 * BK_XPIECES * 1024;   // bytes of one buffer (28 672)
constexpr int BK_TP = 34;                  // row pitch of the intermediate image (pixels)
constexpr int BK_SLAB = 4096;
constexpr size_t BK_LDS = 2 * BK_BUF + BK_R * BK_SLAB;   // 81 920 = 80 KB: two workgroups per CU

// Step structure of a tile.  Units: [U0 | U1 | U2 | U3] of 9 steps each; FIRST (the first block of the network, conv1 on <= 16 input
// channels): [P | U2 | U3] with P = five two-tap steps on the one 16-channel k-slot (k_conv3.h UK_P5), 23 steps and 23 slabs per tile.
// A unit that FETCHES (U0: x chunk 1 of this tile -> buffer 1; U3: x chunk 0 of the next tile -> buffer 0; FIRST has one x chunk, so only
// U3 fetches) issues its 7 halo pieces per wave as 2, 2, 1, 1, 1 at ordinals 0..4 -- all of them at least two steps before the wait
// at ordinal 7 that must cover them.
constexpr int bk_nsteps(bool first) { return first ? 23 : 36; }
constexpr int bk_unit(bool first, int g) { return first ? (g < 5 ? 0 : g < 14 ? 2 : 3) : g / 9; }           // 0..3 (FIRST: 0 = the P unit)
constexpr int bk_ord(bool first, int g) { return first ? (g < 5 ? g : (g - 5) % 9) : g % 9; }
constexpr bool bk_fetches(bool first, int g) { return bk_unit(first, g) == 3 || (!first && bk_unit(first, g) == 0); }
constexpr int bk_pieces_u(int s) { return s <= 1 ? 2 : s <= 4 ? 1 : 0; }
constexpr int bk_first_u(int s) { int n = 0; for (int t = 0; t < s; ++t) n += bk_pieces_u(t); return n; }
constexpr int bk_pieces(bool first, int g) { return bk_fetches(first, g) ? bk_pieces_u(bk_ord(first, g)) : 0; }   // g = step of the tile (cyclic)
// DMA instructions a wave issued after "its part of slab g + 2" as seen at the wait of step g (k_conv3.h c3_younger; IPW = 1)
constexpr int bk_younger(bool first, int g) {
  const int G = bk_nsteps(first);
  int n = BK_R - 3;
  for (int d = 1; d <= BK_R - 2; ++d) n += bk_pieces(first, ((g - d) % G + G) % G);
  const int cap = 9 - 3 - 4;               // ordinal 7 of a fetching unit: the wait also covers the unit's last halo piece (ordinal 4)
  return (bk_fetches(first, g) && bk_ord(first, g) == 7 && n > cap) ? cap : n;
}

// M16 (round 6): the MFMAs as v_mfma_f32_16x16x32 (k_conv3.h M16 says why: the part sustains more of them under its power cap; this kernel
// runs at the cap too, tools/clock_probe.py).  A step is ONE 32-channel k-step: conv1's three flat 32-pixel fragments are 6 half-fragments
// (16 flat positions x 32 channels: lane (n, q) reads slot q), conv2's two rows 4; the 4-KB slab is 4 weight fragments of 16 channels x 32
// (pack_taps_frag16): block 0 = fragments 0, 1, block 1 = 2, 3, pixel-major, half-fragments single-buffered (re-read for the next step right
// after their last MFMA).  Same sums in the same order per accumulator: results bit-identical to the 32 x 32 x 16 form and to the two launches.
template <int DT, bool SC, bool FIRST = false, bool M16 = false>
__global__ __launch_bounds__(256, 2) void block64_stream_kernel(BlockArgs a) {
  static_assert(!FIRST || SC, "the first block changes the channel count: projection shortcut");
  static_assert(!(FIRST && M16), "the first block (five two-tap steps on one 16-channel slot) stays in the 32 x 32 x 16 form");
  HIP_DYNAMIC_SHARED(unsigned char, smem);
  constexpr int R = BK_R, SLAB = BK_SLAB, RING = 2 * BK_BUF, NCT = 2;
  constexpr int ROWB2 = BK_TP * 64;        // bytes of one row of the intermediate image
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, hi = lane >> 5;

  // BatchNorm shifts as rank-1 MFMA operands (k_conv3.h FOLD): per channel block j the dword {hi, lo} of this lane's shift
  unsigned bz1[NCT], bz2[NCT];
#pragma unroll
  for (int j = 0; j < NCT; ++j) {
    const float t1 = a.shift1[32 * j + conv_row_perm(m)], t2 = a.shift2[32 * j + conv_row_perm(m)];
    const bf16_t h1 = H16<DT>::from_f32(t1), h2 = H16<DT>::from_f32(t2);
    const bf16_t l1 = H16<DT>::from_f32(t1 - H16<DT>::to_f32(h1)), l2 = H16<DT>::from_f32(t2 - H16<DT>::to_f32(h2));
    bz1[j] = hi ? 0u : ((unsigned)h1 | ((unsigned)l1 << 16));
    bz2[j] = hi ? 0u : ((unsigned)h2 | ((unsigned)l2 << 16));
  }

  // M16: the shift of MFMA row lane & 15 of 16-channel fragment cb = lane quad (4 fragments: one register per conv)
  unsigned bz1m = 0u, bz2m = 0u;
  if constexpr (M16) {
    const float t1 = a.shift1[conv_row16(lane >> 4, lane & 15)], t2 = a.shift2[conv_row16(lane >> 4, lane & 15)];
    const bf16_t h1 = H16<DT>::from_f32(t1), h2 = H16<DT>::from_f32(t2);
    const bf16_t l1 = H16<DT>::from_f32(t1 - H16<DT>::to_f32(h1)), l2 = H16<DT>::from_f32(t2 - H16<DT>::to_f32(h2));
    bz1m = (unsigned)h1 | ((unsigned)l1 << 16);
    bz2m = (unsigned)h2 | ((unsigned)l2 << 16);
  }

  const int G = gridDim.x, wg = blockIdx.x;
  const int ntl = (a.ntiles - wg + G - 1) / G;             // tiles of this workgroup: wg, wg + G, ...
  const int tiles_img = a.ncol * a.nrow;

  // ---- DMA issue (k_conv3.h: LDS address in M0, inline asm so that hipcc's waitcnt pass does not see the loads) -----------------
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

  // ---- tile cursors (k_conv3.h: decoded once, then advanced by G in mixed radix; XCD-aware order when a.xcd) ---------------------
  const int xdq = (G >> 3) / a.nrow;
  const int g_ct = a.xcd ? (8 * xdq) % a.ncol : G % a.ncol, g_rb = a.xcd ? (G >> 3) % a.nrow : (G / a.ncol) % a.nrow,
            g_b = a.xcd ? (8 * xdq) / a.ncol : G / tiles_img;
  const int g_ct1 = (8 * xdq + 8) % a.ncol, g_b1 = (8 * xdq + 8) / a.ncol;
  auto tile_advance = [&](int& ct, int& rb, int& b) __attribute__((always_inline)) {
    int& d0 = a.xcd ? rb : ct;
    int& d1 = a.xcd ? ct : rb;
    const int r0 = a.xcd ? a.nrow : a.ncol, r1 = a.xcd ? a.ncol : a.nrow;
    d0 += a.xcd ? g_rb : g_ct;
    const int c1 = d0 >= r0 ? 1 : 0;
    d0 -= c1 ? r0 : 0;
    d1 += a.xcd ? (c1 ? g_ct1 : g_ct) : g_rb + c1;
    const int c2 = d1 >= r1 ? 1 : 0;
    d1 -= c2 ? r1 : 0;
    b += (a.xcd && c1 ? g_b1 : g_b) + c2;
  };
  const int xs0 = ((wg >> 3) / a.nrow) * 8 + (wg & 7);
  int f_ct = a.xcd ? xs0 % a.ncol : wg % a.ncol, f_rb = a.xcd ? (wg >> 3) % a.nrow : (wg / a.ncol) % a.nrow,
      f_b = a.xcd ? xs0 / a.ncol : wg / tiles_img;                 // tile of the NEXT x chunk to fetch
  int c_ct = f_ct, c_rb = f_rb, c_b = f_b;                         // tile being computed

  // ---- x fetch: chunk hc of fetch tile -> buffer hc.  Image pixel of flat position p = 36 r + c: (h0 - 2 + r, w0 - 2 + c). ------
  int hk = 0, hc = 0;                                              // (tile ordinal, chunk) of the NEXT fetch unit
  int hh0 = 0, hw0 = 0;
  const unsigned char* htile = nullptr;                            // uniform: image pixel (hh0, hw0), channel 0
  const unsigned char* hbase = nullptr;                            // ... + this unit's chunk
  bool hnew = true;
  auto halo_begin = [&]() {
    if (hnew) {
      hh0 = f_rb * 8 - 2; hw0 = f_ct * 32 - 2;
      htile = (const unsigned char*)(a.x + (size_t)f_b * a.x_bs + a.x_co) + ((long)hh0 * a.W + hw0) * (long)a.x_cs * 2;
      hnew = false;
    }
    hbase = htile + hc * 64;
    // advance: chunk 1 of the same tile, then chunk 0 of the next tile (FIRST: one chunk per tile); past the end of the list the last
    // unit is fetched again
    if (!FIRST && hc == 0) hc = 1;
    else if (hk + 1 < ntl) { ++hk; hc = 0; tile_advance(f_ct, f_rb, f_b); hnew = true; }
  };
  auto halo_piece = [&](int buf, int j) {
    const int q = wave * BK_XPIECES + j;
    int ol = lane;
    asm volatile("" : "+v"(ol));      // (opaque: keeps the per-piece index math out of the registers that live across the MFMA phase)
    const int pp = 16 * q + (ol >> 2), r = (pp * 1821) >> 16, cc = pp - BK_XP * r;      // pp / 36 exactly for pp < 448
    const int hs = (ol & 3) ^ ((ol >> 4) & 3);                      // logical 16-byte slot: physical slot ^ ((pp >> 2) & 3)
    const bool ok = pp < 12 * BK_XP && (unsigned)(hh0 + r) < (unsigned)a.H && (unsigned)(hw0 + cc) < (unsigned)a.W && (!FIRST || hs < a.nslots1);
    const unsigned char* src = hbase + (long)((r * a.W + cc) * a.x_cs * 2 + hs * 16);
    dma_v(ok ? (const void*)src : (const void*)a.zero16, buf + q * 1024);
  };
  int fslot = 0, fslab = 0;
  auto slab_piece = [&]() { dma_s(a.w + (size_t)fslab * SLAB + wave * 1024, lane * 16, RING + fslot * SLAB + wave * 1024); };
  auto slab_advance = [&]() {
    fslot = fslot + 1 == R ? 0 : fslot + 1;
    fslab = fslab + 1 == bk_nsteps(FIRST) ? 0 : fslab + 1;
  };

  // ---- fragment addressing -------------------------------------------------------------------------------------------------------
  // conv1 (flat): fragment i of the wave, tap (dh, dw), k-step ks: position P = 96 w + 32 i + m + 36 dh + dw of the x image,
  //   byte = 64 P + (((2 ks + hi) ^ (P >> 2)) & 3) * 16  -> a1[dh][dw] for i = 0, ks = 0; + 2048 i; ^ 32 for ks = 1
  // conv2 (rows, by-column swizzle): output row 2 w + i, column m, tap (dh, dw): pixel (2 w + i + dh, m + dw) of the t image,
  //   byte = 64 (34 row + col) + (((2 ks + hi) ^ (col >> 2)) & 3) * 16 -> a2[dw] + (i + dh) * ROWB2
  int a1[3][3], a2[3];
#pragma unroll
  for (int dh = 0; dh < 3; ++dh)
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
      const int P = 96 * wave + m + BK_XP * dh + dw;
      a1[dh][dw] = P * 64 + (((hi ^ (P >> 2)) & 3) << 4);
    }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int c = d + m;
    a2[d] = c * 64 + (((hi ^ (c >> 2)) & 3) << 4) + wave * 2 * ROWB2;
  }
  if constexpr (M16) {   // lane (n, q): slot q (all 32 channels of the chunk in one k-step) of flat position / pixel n; +16 positions = +1024 B
    const int n16 = lane & 15, q16 = lane >> 4;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int P = 96 * wave + n16 + BK_XP * dh + dw;
        a1[dh][dw] = P * 64 + (((q16 ^ (P >> 2)) & 3) << 4);
      }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int c = d + n16;
      a2[d] = c * 64 + (((q16 ^ (c >> 2)) & 3) << 4) + wave * 2 * ROWB2;
    }
  }
  const int boff = RING + lane * 16;

  f32x16 acc[M16 ? 1 : 3][M16 ? 1 : NCT];
  f32x4 acc6[M16 ? 6 : 1][M16 ? 4 : 1];      // M16: [pixel half-fragment][16-channel fragment]
  s16x8 fa[2][3], fb[2][NCT];
  s16x8 fa16[6];
#define BK_FENCE() __builtin_amdgcn_sched_barrier(0)
  // fragment read k of a k-step with F pixel fragments: fa[0], fb[0], fb[1], fa[1], (fa[2]); FS = byte stride between pixel fragments
#define BK_RD(F, FS, BUF, K, AADDR, BADDR, KS)                                                              \
  {                                                                                                         \
    if ((K) == 0) fa[BUF][0] = *(const s16x8*)(smem + (AADDR));                                              \
    else if ((K) <= NCT) fb[BUF][(K) - 1] = *(const s16x8*)(smem + (BADDR) + ((KS) * NCT + (K) - 1) * 1024);  \
    else fa[BUF][(K) - NCT] = *(const s16x8*)(smem + (AADDR) + ((K) - NCT) * (FS));                          \
    BK_FENCE();                                                                                             \
  }
#define BK_MM(BUF, N)                                                                                       \
  {                                                                                                         \
    acc[(N) / NCT][(N) % NCT] = H16<DT>::mfma(fb[BUF][(N) % NCT], fa[BUF][(N) / NCT], acc[(N) / NCT][(N) % NCT]); \
    BK_FENCE();                                                                                             \
  }
#define BK_SYNC(VMCNT, LGKM)                                      \
  {                                                               \
    asm volatile("" ::: "memory");                                \
    __builtin_amdgcn_s_waitcnt(C3_WAIT_IMM(VMCNT, LGKM));         \
    __builtin_amdgcn_s_barrier();                                 \
    asm volatile("" ::: "memory");                                \
    BK_FENCE();                                                   \
  }
  // One step of the tile, G_ = its index in the tile (bk_unit / bk_ord), with F pixel fragments per wave:
  //   block 0: MFMAs of k-step 0, the reads of (this step, k-step 1) interleaved 1:1 -- from A1_;
  //   block 1: MFMAs of k-step 1, first half with the reads of (next step, k-step 0) -- from ANEXT_, NF_ pixel fragments at stride NFS_:
  //            the next step's shape (NF_ = 0: none, the pipeline is cut) --, the step's barrier, then the second half with the step's
  //            DMA pieces (the slab, and the halo pieces of a fetching unit into buffer HB_).
#define BK_STEP(F, FS, G_, HB_, NF_, NFS_, A1_, ANEXT_)                                                            \
  {                                                                                                                \
    constexpr int NM_ = (F) * NCT, NR_ = (F) + NCT, NRN_ = (NF_) ? (NF_) + NCT : 0;                                \
    constexpr int NH_ = bk_pieces(FIRST, G_), NP_ = 1 + NH_, HF_ = bk_first_u(bk_ord(FIRST, G_)), YG_ = bk_younger(FIRST, G_); \
    const int acur_ = (A1_);                                                                                       \
    const int bcur_ = boff + rslot * SLAB;                                                                         \
    const int rnext_ = rslot + 1 == R ? 0 : rslot + 1;                                                             \
    const int anext_ = (ANEXT_);                                                                                   \
    const int bnext_ = boff + rnext_ * SLAB;                                                                       \
    BK_FENCE();                                                                                                    \
    _Pragma("unroll") for (int n = 0; n < NM_; ++n) {                                                              \
      BK_MM(0, n)                                                                                                  \
      if (n < NR_) BK_RD(F, FS, 1, n, acur_, bcur_, 1)                                                             \
    }                                                                                                              \
    _Pragma("unroll") for (int n = 0; n < NM_ / 2; ++n) {                                                          \
      BK_MM(1, n)                                                                                                  \
      if (n < NRN_) BK_RD(NF_, NFS_, 0, n, anext_, bnext_, 0)                                                      \
    }                                                                                                              \
    { _Pragma("unroll") for (int n = NM_ / 2; n < NRN_; ++n) BK_RD(NF_, NFS_, 0, n, anext_, bnext_, 0) }           \
    BK_SYNC(YG_, NRN_)                                                                                             \
    if (NH_ > 0 && bk_ord(FIRST, G_) == 0) halo_begin();                                                           \
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
  // conv1 step: unit U_ (0 or 1) reads x chunk U_ from buffer U_; tap S_ (k-step 1 = the same pixels' second 16-channel slot: ^ 32); the
  // next step is tap S_ + 1 of the same buffer, or tap 0 of buffer 1 after U0's last step; U1's last step pre-reads nothing (t does
  // not exist yet).  U0 fetches x chunk 1 into buffer 1.
#define BK_C1(U_, S_)                                                                                              \
  BK_STEP(3, 2048, 9 * (U_) + (S_), BK_BUF, (((U_) == 1 && (S_) == 8) ? 0 : 3), 2048,                              \
          (a1[(S_) / 3][(S_) % 3] + (U_) * BK_BUF) ^ 32,                                                           \
          ((S_) == 8 ? a1[0][0] + BK_BUF : a1[(((S_) + 1) % 9) / 3][((S_) + 1) % 3] + (U_) * BK_BUF))
  // FIRST: conv1 step S_ of the five two-tap steps on x's one 16-channel k-slot: k-step 0 = tap 2 S_, k-step 1 = tap 2 S_ + 1 (tap 9 does
  // not exist: its packed weights are zero, any address will do); the last one pre-reads nothing
#define BK_P1(S_)                                                                                                  \
  BK_STEP(3, 2048, (S_), 0, ((S_) == 4 ? 0 : 3), 2048, a1[((2 * (S_) + 1) % 9) / 3][(2 * (S_) + 1) % 3],           \
          a1[((2 * (S_) + 2) % 9) / 3][(2 * (S_) + 2) % 3])
  // conv2 step: unit 2 + C_ reads t chunk C_ from buffer C_; U3 fetches the next tile's x chunk 0 into buffer 0, and its last step
  // pre-reads the first fragments of the NEXT tile's conv1 (that chunk has landed: the wait of U3's ordinal 7 covered it), so only the
  // cut before conv2 leaves a bubble
#define BK_C2(C_, S_)                                                                                              \
  BK_STEP(2, ROWB2, (FIRST ? 5 : 18) + 9 * (C_) + (S_), 0, (((C_) == 1 && (S_) == 8) ? 3 : 2), (((C_) == 1 && (S_) == 8) ? 2048 : ROWB2), \
          (a2[(S_) % 3] + ((S_) / 3) * ROWB2 + (C_) * BK_BUF) ^ 32,                                                \
          ((S_) == 8 ? ((C_) == 1 ? a1[0][0] : a2[0] + BK_BUF) : a2[((S_) + 1) % 3] + ((((S_) + 1) % 9) / 3) * ROWB2 + (C_) * BK_BUF))

  // ---- M16 forms of the above ---------------------------------------------------------------------------------------------------------
  // half-fragment HX of the step at AADDR: 32-pixel fragment HX >> 1 (stride FS), half HX & 1; weight fragment C of the slab half at BADDR;
  // MFMA N of block BK: half-fragment N >> 1 against weight fragment 2 * BK + (N & 1)
#define BK_RDW16(BUF, C, BADDR) { fb[BUF][C] = *(const s16x8*)(smem + (BADDR) + (C) * 1024); BK_FENCE(); }
#define BK_RDP16(HX, AADDR, FS) { fa16[HX] = *(const s16x8*)(smem + (AADDR) + ((HX) >> 1) * (FS) + ((HX) & 1) * 1024); BK_FENCE(); }
#define BK_MM16(BK, N)                                                                                             \
  {                                                                                                                \
    acc6[(N) >> 1][2 * (BK) + ((N) & 1)] = H16<DT>::mfma16(fb[BK][(N) & 1], fa16[(N) >> 1], acc6[(N) >> 1][2 * (BK) + ((N) & 1)]); \
    BK_FENCE();                                                                                                    \
  }
  // One step with F half-fragments (6: conv1, 4: conv2); the next step has NF_ (0: the pipeline is cut) at fragment stride NFS_ from ANEXT_.
  // Half-fragment hx is re-read right after MFMA 2 hx + 1 of block 1, its last use; half-fragments the next step has more (conv2 -> the next
  // tile's conv1) go into registers this step does not use.  LG_ = the next step's reads issued before the barrier.
#define BK_STEP16(F, G_, HB_, NF_, NFS_, ANEXT_)                                                                   \
  {                                                                                                                \
    constexpr int NM_ = 2 * (F);                                                                                   \
    constexpr int NX_ = (NF_) > (F) ? (NF_) - (F) : 0;                                                             \
    constexpr int LG_ = (NF_) ? 2 + NX_ + ((NF_) < (F) / 2 ? (NF_) : (F) / 2) : 0;                                 \
    constexpr int NH_ = bk_pieces(FIRST, G_), NP_ = 1 + NH_, HF_ = bk_first_u(bk_ord(FIRST, G_)), YG_ = bk_younger(FIRST, G_); \
    const int bcur_ = boff + rslot * SLAB;                                                                         \
    const int rnext_ = rslot + 1 == R ? 0 : rslot + 1;                                                             \
    const int anext_ = (ANEXT_);                                                                                   \
    const int bnext_ = boff + rnext_ * SLAB;                                                                       \
    BK_FENCE();                                                                                                    \
    _Pragma("unroll") for (int n = 0; n < NM_; ++n) {                                                              \
      BK_MM16(0, n)                                                                                                \
      if (n < 2) BK_RDW16(1, n, bcur_ + 2048)                                                                      \
    }                                                                                                              \
    _Pragma("unroll") for (int n = 0; n < NM_ / 2; ++n) {                                                          \
      BK_MM16(1, n)                                                                                                \
      if ((NF_) && n < 2) BK_RDW16(0, n, bnext_)                                                                   \
      if (n < NX_) BK_RDP16((F) + n, anext_, NFS_)                                                                 \
      if ((n & 1) && (n >> 1) < (NF_)) BK_RDP16(n >> 1, anext_, NFS_)                                              \
    }                                                                                                              \
    BK_SYNC(YG_, LG_)                                                                                              \
    if (NH_ > 0 && bk_ord(FIRST, G_) == 0) halo_begin();                                                           \
    _Pragma("unroll") for (int n = NM_ / 2; n < NM_; ++n) {                                                        \
      BK_MM16(1, n)                                                                                                \
      _Pragma("unroll") for (int p = 0; p < NP_; ++p)                                                              \
        if (p * (NM_ / 2) / NP_ == n - NM_ / 2) {                                                                  \
          if (p == 0) slab_piece(); else halo_piece(HB_, HF_ + p - 1);                                             \
          BK_FENCE();                                                                                              \
        }                                                                                                          \
      if ((n & 1) && (n >> 1) < (NF_) && (n >> 1) < (F)) BK_RDP16(n >> 1, anext_, NFS_)                            \
    }                                                                                                              \
    slab_advance();                                                                                                \
    rslot = rnext_;                                                                                                \
  }
#define BK_C1M(U_, S_)                                                                                             \
  BK_STEP16(6, 9 * (U_) + (S_), BK_BUF, (((U_) == 1 && (S_) == 8) ? 0 : 6), 2048,                                  \
            ((S_) == 8 ? a1[0][0] + BK_BUF : a1[(((S_) + 1) % 9) / 3][((S_) + 1) % 3] + (U_) * BK_BUF))
#define BK_C2M(C_, S_)                                                                                             \
  BK_STEP16(4, 18 + 9 * (C_) + (S_), 0, (((C_) == 1 && (S_) == 8) ? 6 : 4), (((C_) == 1 && (S_) == 8) ? 2048 : ROWB2), \
            ((S_) == 8 ? ((C_) == 1 ? a1[0][0] : a2[0] + BK_BUF) : a2[((S_) + 1) % 3] + ((((S_) + 1) % 9) / 3) * ROWB2 + (C_) * BK_BUF))

  // ---- prologue: x chunk 0 of the first tile, a full ring ----------------------------------------------------------------------------
  halo_begin();
#pragma unroll
  for (int j = 0; j < BK_XPIECES; ++j) halo_piece(0, j);
#pragma unroll 1
  for (int s0 = 0; s0 < R; ++s0) { slab_piece(); slab_advance(); }
  BK_SYNC(0, 0)
  int rslot = 0;
  if constexpr (M16) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) BK_RDW16(0, kk, boff + rslot * SLAB)
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) BK_RDP16(kk, a1[0][0], 2048)
  } else {
#pragma unroll
  for (int kk = 0; kk < 3 + NCT; ++kk) BK_RD(3, 2048, 0, kk, a1[0][0], boff + rslot * SLAB, 0)      // first fragments of the first tile
  }

  for (int k = 0; k < ntl; ++k) {
    if constexpr (M16) {
      // ================= the tile in the 16 x 16 x 32 form =========================================================================
      unsigned z0 = 0u, b1 = bz1m, b2 = bz2m;
      int oq = lane >> 4;
      asm volatile("" : "+v"(z0), "+v"(b1), "+v"(b2), "+v"(oq));      // (opaque: no operand tuple may be loop invariant, k_conv3.h)
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      // accumulators of NH half-fragments from the shift: one rank-1 MFMA per 16-channel fragment (A = {hi, lo} in the two k of lane quad cb,
      // B = ones there), the same for every pixel -> register copies for the other half-fragments
      auto init_acc = [&](unsigned bz, int NH) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
          unsigned ob[4] = {oq == cb ? H16<DT>::ONE * 0x10001u : z0, z0, z0, z0};
          unsigned ab[4] = {bz, z0, z0, z0};
          s16x8 ones, bzv;
          memcpy(&ones, ob, 16);
          memcpy(&bzv, ab, 16);
          acc6[0][cb] = H16<DT>::mfma16(bzv, ones, f32x4{});
          BK_FENCE();
        }
#pragma unroll
        for (int h = 1; h < 6; ++h)
          if (h < NH) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc6[h][cb] = acc6[0][cb];
          }
        BK_FENCE();
      };
      // ---- conv1 on the wave's six flat half-fragments ----------------------------------------------------------------------------------
      init_acc(b1, 6);
      BK_C1M(0, 0) BK_C1M(0, 1) BK_C1M(0, 2) BK_C1M(0, 3) BK_C1M(0, 4) BK_C1M(0, 5) BK_C1M(0, 6) BK_C1M(0, 7) BK_C1M(0, 8)
      BK_C1M(1, 0) BK_C1M(1, 1) BK_C1M(1, 2) BK_C1M(1, 3) BK_C1M(1, 4) BK_C1M(1, 5) BK_C1M(1, 6) BK_C1M(1, 7) BK_C1M(1, 8)
      // ---- t = relu(conv1 + shift1), rounded, into buf0 / buf1 in conv2's halo layout: lane (n, q) of accumulators (h, 2 j), (h, 2 j + 1)
      // holds channels 32 j + 8 q .. + 7 of flat position J = 96 w + 16 h + n -- one 16-byte slot (slot q of chunk j)
      const int ct = c_ct, rb = c_rb, b = c_b;
      tile_advance(c_ct, c_rb, c_b);
      {
        int el = lane;
        asm volatile("" : "+v"(el));
        const int en = el & 15, eq = el >> 4;
#pragma unroll
        for (int h = 0; h < 6; ++h) {
          const int J = 96 * wave + 16 * h + en, rr = (J * 1821) >> 16, cc = J - BK_XP * rr;
          const bool keep = cc < BK_TP && rr < 10;
          const bool inside = (unsigned)(rb * 8 - 1 + rr) < (unsigned)a.H && (unsigned)(ct * 32 - 1 + cc) < (unsigned)a.W;
          const int doff = (rr * BK_TP + cc) * 64 + (((eq ^ (cc >> 2)) & 3) << 4);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            BK_FENCE();
            unsigned pk[4];
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
              const f32x4 av = acc6[h][2 * j + (w2 >> 1)];
              unsigned p2 = H16<DT>::pk(av[2 * (w2 & 1)], av[2 * (w2 & 1) + 1]);
              p2 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p2), (s16x2){0, 0}));
              pk[w2] = inside ? p2 : 0u;
            }
            if (keep) *(Slot16*)(smem + j * BK_BUF + doff) = Slot16{pk[0], pk[1], pk[2], pk[3]};
          }
        }
      }
      // ---- conv2 on the wave's two rows (four half-fragments) ---------------------------------------------------------------------------
      init_acc(b2, 4);
      BK_SYNC(63, 0)       // every wave's part of t is in LDS
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) BK_RDW16(0, kk, boff + rslot * SLAB)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) BK_RDP16(kk, a2[0], ROWB2)
      BK_C2M(0, 0) BK_C2M(0, 1) BK_C2M(0, 2) BK_C2M(0, 3) BK_C2M(0, 4) BK_C2M(0, 5) BK_C2M(0, 6) BK_C2M(0, 7) BK_C2M(0, 8)
      BK_C2M(1, 0) BK_C2M(1, 1) BK_C2M(1, 2) BK_C2M(1, 3) BK_C2M(1, 4) BK_C2M(1, 5) BK_C2M(1, 6) BK_C2M(1, 7) BK_C2M(1, 8)
      // ---- epilogue: (+ projection shortcut on the accumulators |) + x, round, ReLU, transpose through this wave's quarter of buf1 ---------
      {
        const int oh0 = rb * 8 + 2 * wave;
        int el = lane;
        asm volatile("" : "+v"(el));
        const int en = el & 15, eq = el >> 4;
        unsigned char* scr = smem + BK_BUF + wave * (BK_BUF / 4);
        bf16_t* __restrict__ yrow0 = a.y + (size_t)b * a.y_bs + (size_t)oh0 * a.W * a.y_cs + a.y_co;
        const bf16_t* __restrict__ rimg0 = a.x + (size_t)b * a.x_bs + a.x_co;
        if constexpr (SC) {
          // acc[px][co] += sum_ci scw[co][ci] * x[px][ci]: B = this wave's output pixels of x from global memory (lane (n, q): channels
          // 32 ks + 8 q .. of pixel 16 half + n), A = the packed 1x1 weights (pack_sc_frag16), two 32-channel k-steps
          const unsigned char* __restrict__ wq = a.scw + el * 16;
          s16x8 sxq[4][2];
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const int ow = ct * 32 + 16 * (h & 1) + en, oh = oh0 + (h >> 1);
            const bool live = ow < a.W && oh < a.H;
            const bf16_t* sp = rimg0 + (live ? ((size_t)oh * a.W + ow) * a.x_cs : 0) + 8 * eq;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) sxq[h][ks] = *(const s16x8*)(sp + 32 * ks);
          }
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
              const s16x8 wf = *(const s16x8*)(wq + (size_t)(ks * 4 + cb) * 1024);
#pragma unroll
              for (int h = 0; h < 4; ++h) acc6[h][cb] = H16<DT>::mfma16(wf, sxq[h][ks], acc6[h][cb]);
            }
          BK_FENCE();
        }
        Slot16 rv[2][2][2];                                  // [buffer][pixel half][fragment pair]
        auto res_load = [&](int i, Slot16 (&dst)[2][2]) {
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const int ow = ct * 32 + 16 * hf + en, oh = oh0 + i;
            const bool live = ow < a.W && oh < a.H;
            const bf16_t* rp = rimg0 + (live ? ((size_t)oh * a.W + (size_t)ow) * a.x_cs : 0) + 8 * eq;
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) dst[hf][p2] = *(const Slot16*)(rp + 32 * p2);
          }
        };
        if constexpr (!SC) res_load(0, rv[0]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if constexpr (!SC) { if (i + 1 < 2) res_load(i + 1, rv[(i + 1) & 1]); }
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
              BK_FENCE();
              unsigned pk[4];
#pragma unroll
              for (int w2 = 0; w2 < 4; ++w2) {
                const f32x4 av = acc6[2 * i + hf][2 * p2 + (w2 >> 1)];
                f32x2 v = {av[2 * (w2 & 1)], av[2 * (w2 & 1) + 1]};
                if constexpr (!SC) v += H16<DT>::unpk(rv[i & 1][hf][p2][w2]);
                const unsigned q2 = H16<DT>::pk(v[0], v[1]);
                pk[w2] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, q2), (s16x2){0, 0}));
              }
              const int px = 16 * hf + en;
              *(Slot16*)(scr + px * 128 + (((4 * p2 + eq) ^ (px & 7)) << 4)) = Slot16{pk[0], pk[1], pk[2], pk[3]};
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
    } else {
    // ================= the tile in the 32 x 32 x 16 form ===============================================================================
    unsigned z0 = 0u;
    asm volatile("" : "+v"(z0));      // (opaque zero: k_conv3.h explains why the operand tuples must not be loop invariant)
    const unsigned one2 = hi ? z0 : H16<DT>::ONE * 0x10001u;
    s16x8 ones;
    {
      unsigned ob[4] = {one2, z0, z0, z0};
      memcpy(&ones, ob, 16);
    }
    // ---- conv1 on the wave's three flat fragments: accumulators start from shift1 ------------------------------------------------
#pragma unroll
    for (int n = 0; n < 3 * NCT; ++n) {
      unsigned ab[4] = {bz1[n % NCT], z0, z0, z0};
      s16x8 bz;
      memcpy(&bz, ab, 16);
      acc[n / NCT][n % NCT] = H16<DT>::mfma(bz, ones, f32x16{});
    }
    BK_FENCE();
    if constexpr (FIRST) {
      BK_P1(0) BK_P1(1) BK_P1(2) BK_P1(3) BK_P1(4)
    } else {
      BK_C1(0, 0) BK_C1(0, 1) BK_C1(0, 2) BK_C1(0, 3) BK_C1(0, 4) BK_C1(0, 5) BK_C1(0, 6) BK_C1(0, 7) BK_C1(0, 8)
      BK_C1(1, 0) BK_C1(1, 1) BK_C1(1, 2) BK_C1(1, 3) BK_C1(1, 4) BK_C1(1, 5) BK_C1(1, 6) BK_C1(1, 7) BK_C1(1, 8)
    }

    // ---- t = relu(conv1 + shift1), rounded, into buf0 (channels 0..31) / buf1 (32..63) in conv2's halo layout ---------------------
    // Every wave is past the barrier of U1's last step, i.e. nobody reads x any more and no DMA targets the two buffers.
    // Lane (m, hi) of accumulator (i, j) holds channels 32 j + 16 hi + 0..15 of flat position J = 96 w + 32 i + m = (r', c').
    const int ct = c_ct, rb = c_rb, b = c_b;
    tile_advance(c_ct, c_rb, c_b);
    {
      int em = m, ehi = hi;
      asm volatile("" : "+v"(em), "+v"(ehi));
      typedef short s16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int J = 96 * wave + 32 * i + em, rr = (J * 1821) >> 16, cc = J - BK_XP * rr;
        const bool keep = cc < BK_TP && rr < 10;                                    // a position conv2 reads
        const bool inside = (unsigned)(rb * 8 - 1 + rr) < (unsigned)a.H && (unsigned)(ct * 32 - 1 + cc) < (unsigned)a.W;
        const int doff = (rr * BK_TP + cc) * 64 + ((((2 * ehi) ^ (cc >> 2)) & 3) << 4);
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
          BK_FENCE();
          unsigned pk[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            unsigned p2 = H16<DT>::pk(acc[i][j][2 * q], acc[i][j][2 * q + 1]);
            p2 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p2), (s16x2){0, 0}));   // ReLU on the rounded pair
            pk[q] = inside ? p2 : 0u;
          }
          if (keep) {
            *(Slot16*)(smem + j * BK_BUF + doff) = Slot16{pk[0], pk[1], pk[2], pk[3]};
            *(Slot16*)(smem + j * BK_BUF + (doff ^ 16)) = Slot16{pk[4], pk[5], pk[6], pk[7]};
          }
        }
      }
    }
    // ---- conv2 on the wave's two rows: accumulators start from shift2; the first fragments are read after the barrier ------------
#pragma unroll
    for (int n = 0; n < 2 * NCT; ++n) {
      unsigned ab[4] = {bz2[n % NCT], z0, z0, z0};
      s16x8 bz;
      memcpy(&bz, ab, 16);
      acc[n / NCT][n % NCT] = H16<DT>::mfma(bz, ones, f32x16{});
    }
    BK_FENCE();
    BK_SYNC(63, 0)       // every wave's part of t is in LDS (no DMA wait: vmcnt 63 = none)
#pragma unroll
    for (int kk = 0; kk < 2 + NCT; ++kk) BK_RD(2, ROWB2, 0, kk, a2[0], boff + rslot * SLAB, 0)
    BK_C2(0, 0) BK_C2(0, 1) BK_C2(0, 2) BK_C2(0, 3) BK_C2(0, 4) BK_C2(0, 5) BK_C2(0, 6) BK_C2(0, 7) BK_C2(0, 8)
    BK_C2(1, 0) BK_C2(1, 1) BK_C2(1, 2) BK_C2(1, 3) BK_C2(1, 4) BK_C2(1, 5) BK_C2(1, 6) BK_C2(1, 7) BK_C2(1, 8)

    // ---- epilogue (k_conv3.h, the RD_ADD | RD_RELU_POST form): + x, round, ReLU, transpose through this wave's quarter of buf1, ----
    // whole pixel rows to global memory with non-temporal stores.  Nobody reads buf1 after the barrier of U3's last step.
    {
      const int oh0 = rb * 8 + 2 * wave;
      int em = m, ehi = hi, el = lane;
      asm volatile("" : "+v"(em), "+v"(ehi), "+v"(el));
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      unsigned char* scr = smem + BK_BUF + wave * (BK_BUF / 4);       // 7 KB >= 32 pixels x 128 B
      bf16_t* __restrict__ yrow0 = a.y + (size_t)b * a.y_bs + (size_t)oh0 * a.W * a.y_cs + a.y_co;
      const bf16_t* __restrict__ rimg0 = a.x + (size_t)b * a.x_bs + a.x_co;
      if constexpr (SC) {
        // projection shortcut (k_conv3.h SC): acc[px][co] += sum_ci scw[co][ci] * x[px][ci] on the conv's own accumulators -- B operand =
        // this wave's output pixels of x straight from global memory (L2: the tile's x was fetched moments ago), A = the packed 1x1
        // weights; four 16-channel k-steps in the unfused launch's order
        const bf16_t* __restrict__ sb = a.x + (size_t)b * a.x_bs + a.x_co + 8 * ehi;
        const unsigned char* __restrict__ wq = a.scw + el * 16;
        constexpr int SNK = FIRST ? 1 : 4;                 // 16-channel k-steps of the block input
        s16x8 sxq[2][SNK];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int ow = ct * 32 + em, oh = oh0 + i;
          const bool live = ow < a.W && oh < a.H;
          const bf16_t* sp = sb + (live ? ((size_t)oh * a.W + ow) * a.x_cs : 0);
#pragma unroll
          for (int ks = 0; ks < SNK; ++ks) sxq[i][ks] = *(const s16x8*)(sp + 16 * ks);
        }
#pragma unroll
        for (int ks = 0; ks < SNK; ++ks)
#pragma unroll
          for (int j = 0; j < NCT; ++j) {
            const s16x8 wf = *(const s16x8*)(wq + (size_t)(ks * NCT + j) * 1024);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = H16<DT>::mfma(wf, sxq[i][ks], acc[i][j]);
          }
        BK_FENCE();
      }
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
      if constexpr (!SC) res_load(0, rv[0]);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if constexpr (!SC) { if (i + 1 < 2) res_load(i + 1, rv[(i + 1) & 1]); }
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
          BK_FENCE();
          const int cb = j * 32 + 16 * ehi;
          unsigned pk[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            f32x2 v = {acc[i][j][2 * q], acc[i][j][2 * q + 1]};
            if constexpr (!SC) v += H16<DT>::unpk(rv[i & 1][j][q >> 2][q & 3]);
            unsigned p2 = H16<DT>::pk(v[0], v[1]);
            pk[q] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p2), (s16x2){0, 0}));
          }
#pragma unroll
          for (int u = 0; u < 2; ++u)
            *(Slot16*)(scr + em * 128 + ((((cb >> 3) + u) ^ (em & 7)) << 4)) = Slot16{pk[4 * u], pk[4 * u + 1], pk[4 * u + 2], pk[4 * u + 3]};
        }
        BK_FENCE();
        __builtin_amdgcn_wave_barrier();
        // read back pixel-major and store: lane -> (pixel it * 8 + el / 8, slot el % 8)
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
    }   // !M16
  }
  __builtin_amdgcn_s_waitcnt(RD_VMCNT_IMM(0));      // the dummy tail fetches target this workgroup's LDS: retire them before it is released
#undef BK_C2M
#undef BK_C1M
#undef BK_STEP16
#undef BK_MM16
#undef BK_RDP16
#undef BK_RDW16
#undef BK_C2
#undef BK_P1
#undef BK_C1
#undef BK_STEP
#undef BK_SYNC
#undef BK_MM
#undef BK_RD
#undef BK_FENCE
}

// packed weights of a fused block: [conv1 | conv2 | zero tail].  conv2: pack_taps_frag(9 taps, 64, 64).  conv1 with cin = 64: the same;
// with cin <= 16 (FIRST): pack_body_frag(body 3: five two-tap steps on one 16-channel k-slot).  w1: (64, cin, 3, 3), w2: (64, 64, 3, 3)
// row-major, scale1 / scale2: the folded BatchNorm scales (nullptr = none)
inline bool block64_first(int cin) { return cin <= 16; }
inline size_t block64_body_bytes(int cin) { return (size_t)bk_nsteps(block64_first(cin)) * BK_SLAB; }
inline size_t block64_packed_bytes(int cin) { return block64_body_bytes(cin) + RD_CONV_TAIL; }
// ... for the M16 form (cin = 64 only): both convs as pack_taps_frag16
inline void pack_block64_m16(const float* w1, const float* s1, const float* w2, const float* s2, int dt, void* out) {
  memset(out, 0, block64_packed_bytes(64));
  pack_taps_frag16(9, 64, 64, out, [&](int co, int ci, int t) { return (s1 ? s1[co] : 1.f) * w1[((size_t)co * 64 + ci) * 9 + t]; }, dt);
  pack_taps_frag16(9, 64, 64, (unsigned char*)out + 18 * BK_SLAB, [&](int co, int ci, int t) { return (s2 ? s2[co] : 1.f) * w2[((size_t)co * 64 + ci) * 9 + t]; }, dt);
}
// the 1x1 projection shortcut 64 -> 64 of the M16 form: [32-channel k-step (2)][16-channel fragment (4)][64 lanes][8], lane (mm, q) holds
// scale[co] * w[co = conv_row16(cb, mm)][ci = 32 ks + 8 q + j]   (8 KB, like pack_sc_frag)
inline void pack_sc_frag16(const float* w, const float* scale, void* out, int dt = RD_BF16) {
  bf16_t* o = (bf16_t*)out;
  for (int ks = 0; ks < 2; ++ks)
    for (int cb = 0; cb < 4; ++cb)
      for (int lane = 0; lane < 64; ++lane) {
        const int co = conv_row16(cb, lane & 15);
        for (int j = 0; j < 8; ++j) {
          const int ci = 32 * ks + 8 * (lane >> 4) + j;
          *o++ = h16_from_f32(dt, (scale ? scale[co] : 1.f) * w[(size_t)co * 64 + ci]);
        }
      }
}
inline void pack_block64(const float* w1, const float* s1, const float* w2, const float* s2, int cin, int dt, void* out) {
  memset(out, 0, block64_packed_bytes(cin));
  if (block64_first(cin))
    pack_body_frag(3, cin, 64, out, [&](int co, int ci, int dh, int dw) { return (s1 ? s1[co] : 1.f) * w1[(((size_t)co * cin + ci) * 3 + dh) * 3 + dw]; }, dt);
  else
    pack_taps_frag(9, 64, 64, out, [&](int co, int ci, int t) { return (s1 ? s1[co] : 1.f) * w1[((size_t)co * 64 + ci) * 9 + t]; }, dt);
  pack_taps_frag(9, 64, 64, (unsigned char*)out + block64_body_bytes(cin) - 18 * BK_SLAB,
                 [&](int co, int ci, int t) { return (s2 ? s2[co] : 1.f) * w2[((size_t)co * 64 + ci) * 9 + t]; }, dt);
}

inline int launch_block64(const void* x, int x_cs, int x_co, int cin, const void* w, const float* shift1, const float* shift2, const void* sc_w,
                          void* y, int y_cs, int y_co, int B, int H, int W, int dt, hipStream_t st, bool m16 = false) {
  RD_REQUIRE(is_h16(dt), RD_EINVAL, "block64: dtype %d (RD_BF16 or RD_F16)", dt);
  const bool first = block64_first(cin);
  RD_REQUIRE(!m16 || !first, RD_ESHAPE, "block64: the 16 x 16 x 32 form takes 64 input channels");
  RD_REQUIRE(cin == 64 || (first && cin >= 1), RD_ESHAPE, "block64: %d input channels (64, or at most 16 for the network's first block)", cin);
  RD_REQUIRE(!first || sc_w, RD_EINVAL, "block64: a block that changes the channel count needs its projection shortcut");
  BlockArgs a;
  memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.x_cs = x_cs; a.x_co = x_co; a.x_bs = (long)H * W * x_cs;
  a.w = (const unsigned char*)w; a.shift1 = shift1; a.shift2 = shift2; a.scw = (const unsigned char*)sc_w;
  a.y = (bf16_t*)y; a.y_cs = y_cs; a.y_co = y_co; a.y_bs = (long)H * W * y_cs;
  a.zero16 = (const unsigned char*)w + block64_body_bytes(cin);
  a.H = H; a.W = W; a.B = B;
  a.nslots1 = std::min(2, (x_cs - x_co) / 8);      // (channels past cin are zero in the buffer and meet zero weights)
  a.ncol = (W + 31) / 32; a.nrow = (H + 7) / 8; a.ntiles = a.ncol * a.nrow * B;
  const int grid = std::min(a.ntiles, conv_num_cus() * 2);
  a.xcd = dev_switches().conv_xcd && (a.ncol * B) % 8 == 0 && grid % 8 == 0;
  ProfScope ps(RD_PROF_BLOCK, st);
  static std::atomic<unsigned long long> seen{0};
  once_per_device(seen, [] {
    allow_big_lds(block64_stream_kernel<RD_F16, false>); allow_big_lds(block64_stream_kernel<RD_BF16, false>);
    allow_big_lds(block64_stream_kernel<RD_F16, true>); allow_big_lds(block64_stream_kernel<RD_BF16, true>);
    allow_big_lds(block64_stream_kernel<RD_F16, true, true>); allow_big_lds(block64_stream_kernel<RD_BF16, true, true>);
    allow_big_lds(block64_stream_kernel<RD_F16, false, false, true>); allow_big_lds(block64_stream_kernel<RD_BF16, false, false, true>);
    allow_big_lds(block64_stream_kernel<RD_F16, true, false, true>); allow_big_lds(block64_stream_kernel<RD_BF16, true, false, true>);
  });
#define BK_GO(DT_)                                                                                                         \
  {                                                                                                                        \
    if (m16 && sc_w) hipLaunchKernelGGL((block64_stream_kernel<DT_, true, false, true>), dim3(grid), dim3(256), BK_LDS, st, a);  \
    else if (m16) hipLaunchKernelGGL((block64_stream_kernel<DT_, false, false, true>), dim3(grid), dim3(256), BK_LDS, st, a);    \
    else if (first) hipLaunchKernelGGL((block64_stream_kernel<DT_, true, true>), dim3(grid), dim3(256), BK_LDS, st, a);    \
    else if (sc_w) hipLaunchKernelGGL((block64_stream_kernel<DT_, true>), dim3(grid), dim3(256), BK_LDS, st, a);           \
    else hipLaunchKernelGGL((block64_stream_kernel<DT_, false>), dim3(grid), dim3(256), BK_LDS, st, a);                    \
  }
  if (dt == RD_F16) BK_GO(RD_F16) else BK_GO(RD_BF16)
#undef BK_GO
  return check_launch("block64_stream_kernel");
}

}  // namespace rd
