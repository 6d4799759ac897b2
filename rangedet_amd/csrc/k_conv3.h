// 3x3 / stride-1 / pad-1 bf16 convolution as a persistent, fully asynchronous MFMA pipeline (the bulk of the
// RangeDet FLOPs: backbone BasicBlocks dla_backbone.py:18-56 and the head towers head/builder.py:221-240).
//
// One workgroup (4 waves, one per SIMD, up to 512 registers each) per CU walks a list of output tiles of
// 8 rows x 62 columns x all Cout.  Wave w owns output rows 2w, 2w+1 of the tile: 4 pixel fragments (2 rows x 2 x 32 px)
// x NCT channel fragments (32 ch) = 4*NCT accumulators of 32x32 (cout 128: all 256 accumulation registers).
//
//   unit  = (tile, 32-channel k-chunk): one halo image 10 rows x 64 columns x 64 B = 40 KB in LDS, DOUBLE buffered:
//           the halo of unit u+1 (possibly the next tile) is fetched by LDS-DMA while unit u is computed;
//   step  = (unit, tap): one weight slab Cout x 32 ch = NCT*2 KB, kept in MFMA-fragment order (a linear copy of the
//           packed global image) in an R-deep LDS ring filled by LDS-DMA R steps ahead;
//   one workgroup barrier per step (= 2 k-steps x 4*NCT MFMAs per wave), placed between the two k-steps so that the
//   LDS reads it retires were issued a whole MFMA block earlier.  Fragments for the next k-step are always read
//   before the MFMA block of the current one (explicit register double buffer), because with one wave per SIMD
//   nothing else hides LDS latency.
//
// All DMA traffic of a wave retires in order, so "my part of slab g+1 has landed" is a counted s_waitcnt vmcnt(N) with
// N = DMA instructions issued after it -- a compile-time constant per tap because every step issues exactly IPW slab
// instructions and every unit 10 halo instructions (dummy re-fetches keep that true at the end of the list).
// LDS bytes per MFMA: (4 + NCT) KB / (4*NCT) = 0.5 KB (cout 128), 0.75 KB (cout 64) -- a quarter of ds_read_b128 peak.
#pragma once
#include "k_conv.h"

namespace rd {

struct Conv3Args {
  const bf16_t* x; int x_cs, x_co; long x_bs;
  const unsigned char* w;
  const float* scale; const float* shift;
  const bf16_t* res; int r_cs, r_co; long r_bs;
  bf16_t* y; int y_cs, y_co; long y_bs;
  const unsigned char* zero16;  // 16 zero bytes in device memory: DMA source of padding pixels / channels
  int H, W, B, nslots, nchunk, flags, ncol, nrow, ntiles;
  int sw, Wo;   // column stride (1 or 2) and output width: a stride-2 conv is the stride-1 conv with only the even columns stored
  // XCD-aware tile order (xcd = 1; launch_conv3 sets it when the column strips divide by 8): workgroups are dealt round-robin to the
  // 8 XCDs, each with its own L2.  In the plain order (column tile fastest) the tile above / below / beside a tile runs on another
  // XCD, so the 2 shared halo rows of every 10 cross the fabric twice.  Here list position v = wg + k*G is read as
  // (XCD x = v % 8, position j = v / 8 on that XCD) and the XCD walks its own column strips (strip s = column tile x image, s % 8 == x)
  // top to bottom: tile = (strip 8*(j / nrow) + x, row block j % nrow).  The 64 workgroups of an XCD work on 8 whole strips at a time.
  int xcd;
  unsigned long long* trace;
  // HEAD variant: the 1x1 output conv that consumes this conv's result is applied in the epilogue and y is never written
  const unsigned char* hw;   // packed head weights: [hi | lo][ks 0..7][64 lanes][8 bf16] = 16 KB (pack_head_frag)
  const float* hb;           // [hn] bias
  float* ho; long ho_bs, ho_off; int hn;   // out[b*ho_bs + (ho_off + h*W + w)*hn + o]
  // SC variant: the block's 1x1 projection shortcut (dla_backbone.py:44-51) is accumulated onto the tile in the epilogue:
  // acc[px][co] += sum_ci scw[co][ci] * sx[px][ci] (same output pixel grid), so neither the shortcut tensor nor a second
  // launch exists.  BN scales are folded into both weight sets by the packer (one affine for the sum: shift only).
  const bf16_t* sx; int s_cs, s_co; long s_bs;
  const unsigned char* scw;  // packed [ks][Cout/32][64 lanes][8 bf16] (pack_sc_frag)
  int s_nks;                 // 16-channel k-steps of the shortcut input (<= 8)
  // GRP variant: TWO problems of the same shape in one launch (the cls and the reg tower conv of a head level,
  // head/builder.py:221-240).  Images B .. 2B-1 of the tile list are problem 1; every per-problem pointer / size of problem 1 is
  // problem 0's plus the delta below (elements of the pointer's type; g_w, g_hw in bytes), and the image index restarts at 0.
  int ngrp;                  // 1 or 2
  long g_x, g_res, g_y, g_w, g_shift, g_hw, g_hb, g_ho, g_ho_bs;
  int g_hn;
  // PH variant (TS == 3): ALL phases of a transposed conv in one launch (dla_backbone.py:117-127, mxnext/simple.py:545-580).  The
  // tile list is (spatial tile, phase) with the phase running fastest inside a workgroup's list, so the nph phases of a tile are
  // computed back to back by one workgroup: its halo comes from HBM once (the re-fetches of the other phases hit L2) instead of
  // once per phase launch, and a small layer (W = 166: 384 tiles on 512 slots) gets nph times the work per resident slot.
  // Phase ph uses weight image w + ph * w_pb, the two-column tap set 1 (dw in {-1, 0}) or 2 (dw in {0, +1}) by bit ph of ts_mask,
  // and writes / reads channels y_co + ph * y_pc (residual: r_co + ph * r_pc) of the output seen as [H][W][nph * C].
  int nph, ts_mask, y_pc, r_pc;
  long w_pb;
  // Two-tensor input (8 x 32 tiles only): the conv runs over the channel concatenation [x (nchunk1 32-channel chunks) | x2] without
  // that tensor ever existing -- dla_backbone.py:153-154 concatenates the 8-channel range image with the 64 agg3 channels, and a
  // shared 80-channel buffer gives the agg3 producer a 160-byte pixel pitch (every 128-byte row it stores straddles two cache
  // lines).  Chunk c >= nchunk1 is chunk c - nchunk1 of x2 (same H, W; its own channel stride / offset / slot count).
  const bf16_t* x2; int x2_cs, x2_co, nchunk1, nslots2; long x2_bs;
};

// Tile = 8 output rows x 62 columns (halo 10 x 64 pixels): wave w owns rows 2w, 2w+1, each as two 32-pixel fragments, so
// pixel fragment i of a wave = row 2w + (i >> 1), columns 32*(i & 1) ..  (+2048 bytes per fragment in the halo image, as
// a halo row is 64 pixels x 64 B = two fragments).  Against 4 x 126 tiles: 10/8 instead of 6/4 halo rows per output row
// (less HBM and LDS-DMA traffic per pixel) and a finer column grid (2656 = 42.8 tiles of 62: 99.6 % of the computed
// columns are real, 95.8 % with 126; 664: 97 % instead of 88 %; 166: 89 % instead of 66 %).
//
// FPW = pixel fragments per wave.  FPW 4 is the tile above (one workgroup per CU, one wave per SIMD with the whole register
// file).  FPW 2: tile = 4 rows x 62 columns (halo 6 x 64), wave w owns row w; half the accumulators (256 registers per wave)
// and 75 KB of LDS, so TWO workgroups share a CU (two waves per SIMD): each is its own asynchronous pipeline with its own
// barriers, and one's barrier waits, DMA-issue stalls and epilogue run under the other's MFMAs.  The price is 6/4 instead
// of 10/8 halo rows per output row, so it is for the MFMA-bound 128-channel layers, not for the HBM-bound 64-channel ones.
//
// FC = 32-pixel fragments per tile row (2: 62 output columns, halo 64; 1: 30 output columns, halo 32).  <FPW 2, FC 1> is the
// 8-row x 30-column tile: wave w owns rows 2w, 2w+1 like the 8 x 62 tile (10/8 halo rows), half as wide, 72 KB of LDS, two
// workgroups per CU -- the 4-row tile's stall hiding without its 6/4 halo rows, for the HBM-bound 64-channel layers.
constexpr int C3_TW = 62;                  // output columns per tile (halo = 64 columns exactly), FC 2
constexpr int C3_ROWB = 64 * 64;           // bytes of one halo row, FC 2
// NHB = halo buffers (round 3).  2: the halo of unit u+1 is fetched while unit u is consumed.  3 (cout 64 on the 8 x 30 tiles
// only -- the HBM-bound layers): the fetch runs TWO units ahead, so a piece has a whole unit (>= 9 steps) to arrive instead of
// the 3 - 8 steps between its issue and the barrier of ordinal NS-2, and twice as many bytes are in flight per CU; the third
// 20-KB buffer is paid for with a 5-deep instead of 8-deep weight ring and the (FOLD-unused) scale / shift array.
// WD (round 3): the 8-row tile with ALL 32 columns of its one fragment per row live -- 8 x 32 outputs from a halo image of 10 rows
// x 34 pixels (row pitch 34 pixels = 2 176 B instead of 32).  On the 8 x 30 tile two of every 32 MFMA columns compute pixels that
// are thrown away (6.25 % of every conv's MFMAs, and the convs are MFMA / power bound); 2656 = 83 x 32 exactly.
// The halo's 340 pixels are 21.25 one-KB pieces: 6 per wave (the last ones fetch the zero page), LDS = 2 x 24 KB + ring = 80 KB
// exactly (no scale / shift array: FOLD only; cout 64 goes back to two halo buffers and its 8-deep ring).
template <int NCT, int FPW = 4, int FC = 2, int NHB = 2, bool WD = false> struct C3Cfg {
  static_assert(!WD || (FPW == 2 && FC == 1 && NHB == 2), "wide tile: 8 x 32, two halo buffers");
  static constexpr int RW = FPW / FC;                    // output rows per wave
  static constexpr int TH = 4 * RW;                      // output rows per tile (8 or 4)
  static constexpr int COLS = WD ? 34 : 32 * FC;         // halo columns (= row pitch of the halo image in pixels)
  static constexpr int TW = WD ? 32 : COLS - 2;          // output columns per tile
  static constexpr int ROWB = COLS * 64;                 // bytes of one halo row
  static constexpr int HROWS = TH + 2;                   // halo rows
  static constexpr int NPX = HROWS * COLS;               // pixels of one halo image
  static constexpr int HPW = WD ? (NPX + 63) / 64 : HROWS * ROWB / 4096;   // 1-KB halo pieces (16 pixels) per wave and unit
  static constexpr int HALO = HPW * 4096;                // bytes of one halo buffer (WD: incl. the padding pieces)
  static constexpr int HPC = WD ? 7 : HPW;               // schedule code of c3_halo_pieces
  static_assert(NHB == 2 || (NHB == 3 && FPW == 2 && FC == 1 && NCT == 2), "three halo buffers: cout 64 on 8 x 30 tiles");
  static constexpr int R = NHB == 3 ? 5 : FPW == 4 ? (NCT == 4 ? 7 : 10) : FC == 1 ? (NCT == 4 ? 4 : 8) : (NCT == 4 ? 3 : 6);   // ring depth (slabs)
  static constexpr int IPW = NCT / 2;                    // slab DMA instructions per wave per step
  static constexpr int SLAB = NCT * 2048;
  static constexpr size_t LDS = (size_t)NHB * HALO + (size_t)R * SLAB + (NHB == 3 || WD ? 0 : 2 * NCT * 32 * sizeof(float));
};
constexpr int C3_TH = C3Cfg<4, 4>::TH;     // (the FPW 4 geometry, for code that sizes things before choosing a variant)
constexpr int C3_HALO = C3Cfg<4, 4>::HALO;

// packed bf16 conv-family weights: [32-ch chunk][tap][ks (2)][Cout/32][64 lanes][8 bf16], lane (mm, hi) of a fragment
// holds W[co = 32*cb + conv_row_perm(mm)][ci = 32*chunk + 16*ks + 8*hi + j][tap].   get(co, ci, tap) -> float
template <class F>
inline void pack_taps_frag(int ntaps, int cin, int cout, void* out, F get, int dt = RD_BF16) {
  const int nchunk = (cin + 31) / 32, ncb = cout / 32;
  bf16_t* o = (bf16_t*)out;
  for (int c = 0; c < nchunk; ++c)
    for (int t = 0; t < ntaps; ++t)
      for (int ks = 0; ks < 2; ++ks)
        for (int cb = 0; cb < ncb; ++cb)
          for (int lane = 0; lane < 64; ++lane) {
            const int co = cb * 32 + conv_row_perm(lane & 31);
            for (int j = 0; j < 8; ++j) {
              const int ci = c * 32 + ks * 16 + (lane >> 5) * 8 + j;
              *o++ = h16_from_f32(dt, ci < cin ? get(co, ci, t) : 0.f);
            }
          }
}

// M16 (round 6): the same weights for the v_mfma_f32_16x16x32 form of the kernel: [32-ch chunk][tap][Cout/16][64 lanes][8], lane (mm, q)
// of fragment cb holds W[co = conv_row16(cb, mm)][ci = 32*chunk + 8*q + j][tap] -- one fragment is 16 output channels x all 32 input
// channels of the chunk (the 32 x 32 x 16 form: 32 output channels x 16 input channels); same bytes, same slab size.
// MFMA row mm = 4*qd + r lands in accumulator register r of the lanes of quad qd (D[m][n]: lane n + 16*(m/4), register m%4), and the
// row -> channel map gives lane quad qd the 8 CONTIGUOUS channels 32*(cb/2) + 8*qd .. +7 from the fragment pair (cb even, cb odd): one
// 16-byte slot of the output pixel per pair.
__host__ __device__ inline int conv_row16(int cb, int mm) { return 32 * (cb >> 1) + 8 * (mm >> 2) + 4 * (cb & 1) + (mm & 3); }
template <class F>
inline void pack_taps_frag16(int ntaps, int cin, int cout, void* out, F get, int dt = RD_BF16) {
  const int nchunk = (cin + 31) / 32, ncb = cout / 16;
  bf16_t* o = (bf16_t*)out;
  for (int c = 0; c < nchunk; ++c)
    for (int t = 0; t < ntaps; ++t)
      for (int cb = 0; cb < ncb; ++cb)
        for (int lane = 0; lane < 64; ++lane) {
          const int co = conv_row16(cb, lane & 15);
          for (int j = 0; j < 8; ++j) {
            const int ci = c * 32 + (lane >> 4) * 8 + j;
            *o++ = h16_from_f32(dt, ci < cin ? get(co, ci, t) : 0.f);
          }
        }
}

// packed 1x1 output-conv weights of the HEAD variant: [hi | lo][ks 0..7][64 lanes][8 bf16]; lane (mm, hi) of k-step ks holds
// w[mm][16*ks + 8*hi + j] (rows mm >= nout and channels >= cin are zero), hi = bf16(w), lo = bf16(w - hi).
inline void pack_head_frag(const float* w, int nout, int cin, void* out, int dt = RD_BF16) {
  bf16_t* o = (bf16_t*)out;
  for (int part = 0; part < 2; ++part)
    for (int ks = 0; ks < 8; ++ks)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int mm = lane & 31, c = 16 * ks + 8 * (lane >> 5) + j;
          const float v = (mm < nout && c < cin) ? w[(size_t)mm * cin + c] : 0.f;
          const bf16_t h = h16_from_f32(dt, v);
          *o++ = part == 0 ? h : h16_from_f32(dt, v - h16_to_f32(dt, h));
        }
}

// ... for the M16 form of the kernel (16 x 16 x 32 MFMAs, 16 output rows: nout <= 8 needs no more): [hi | lo][pair p 0..3][64 lanes][8],
// lane (mm, q) of fragment p holds w[mm][32*p + 8*q + j] -- the 8 channels lane quad q of the conv's epilogue holds for fragment pair p of
// its pixel, in order, so the rounded accumulators ARE the B operand (no trip through LDS): 2 x 4 KB
inline void pack_head_frag16(const float* w, int nout, int cin, void* out, int dt = RD_BF16) {
  bf16_t* o = (bf16_t*)out;
  for (int part = 0; part < 2; ++part)
    for (int p = 0; p < 4; ++p)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int mm = lane & 15, c = 32 * p + 8 * (lane >> 4) + j;
          const float v = (mm < nout && c < cin) ? w[(size_t)mm * cin + c] : 0.f;
          const bf16_t h = h16_from_f32(dt, v);
          *o++ = part == 0 ? h : h16_from_f32(dt, v - h16_to_f32(dt, h));
        }
}

// packed 1x1 projection-shortcut weights of the SC variant: [ks][Cout/32][64 lanes][8 bf16], lane (mm, hi) of fragment
// (ks, cb) holds scale[co] * w[co = 32*cb + conv_row_perm(mm)][ci = 16*ks + 8*hi + j]  (w: (cout, cin) row-major).
inline size_t sc_frag_bytes(int cin, int cout) { return (size_t)((cin + 15) / 16) * (cout / 32) * 1024; }
inline void pack_sc_frag(const float* w, const float* scale, int cin, int cout, void* out, int dt = RD_BF16) {
  bf16_t* o = (bf16_t*)out;
  const int nks = (cin + 15) / 16;
  for (int ks = 0; ks < nks; ++ks)
    for (int cb = 0; cb < cout / 32; ++cb)
      for (int lane = 0; lane < 64; ++lane) {
        const int co = cb * 32 + conv_row_perm(lane & 31);
        for (int j = 0; j < 8; ++j) {
          const int ci = ks * 16 + (lane >> 5) * 8 + j;
          *o++ = h16_from_f32(dt, ci < cin ? (scale ? scale[co] : 1.f) * w[(size_t)co * cin + ci] : 0.f);
        }
      }
}

// Tap sets.  TS = 0: all nine taps (convs).  A transposed-conv phase only has taps in two of the three columns:
// TS = 1 -> dw in {-1, 0}, TS = 2 -> dw in {0, +1}; its unit is 6 steps instead of 9 (no MFMAs on zero weights).
constexpr int c3_nsteps(int TS) { return TS == 0 ? 9 : 6; }
// TS = 3: a two-column tap set whose side (1 or 2) is a run-time property of the tile (Conv3Args::ts_mask): numbered like TS 1, the
// column index T % 3 in {0, 1} then selects one of the tile's two column offsets instead of a fixed one
constexpr int c3_tap(int TS, int s) {            // tap index T = 3*(dh+1) + (dw+1) of step ordinal s
  return TS == 0 ? s : 3 * (s / 2) + (s % 2) + (TS == 2 ? 1 : 0);
}
// Halo pieces (C3_HPW = 10 per wave and unit) per step ordinal: 2 each at ordinals 0..4 of a 9-step unit; 4, 4, 2 at
// ordinals 0..2 of a 6-step unit; the last ones are issued at least two steps before the wait that must cover them
// (ordinal NS-2).
// (HP = 6, the 4-row tile: 2 each at ordinals 0..2 of a 9-step unit, 3 each at ordinals 0..1 of a 6-step unit;
//  HP = 5, the 8 x 30 tile: 1 each at ordinals 0..4 of a 9-step unit; 2, 2, 1 at ordinals 0..2 of a 6-step unit.)
// (HP = 7 is the code of the wide 8 x 32 tile: 6 pieces, 2, 1, 1, 1, 1 at ordinals 0..4 of a 9-step unit; 2 each at 0..2 of a 6-step unit.)
constexpr int c3_halo_last(int NS, int HP) { return HP == 7 ? (NS == 9 ? 4 : 2) : HP == 6 ? (NS == 9 ? 2 : 1) : (NS == 9 ? 4 : 2); }
constexpr int c3_halo_pieces(int s, int NS, int HP) {
  return HP == 7 ? (NS == 9 ? (s == 0 ? 2 : s <= 4 ? 1 : 0) : (s <= 2 ? 2 : 0))
       : HP == 10 ? (NS == 9 ? (s <= 4 ? 2 : 0) : (s <= 1 ? 4 : (s == 2 ? 2 : 0)))
       : HP == 6  ? (NS == 9 ? (s <= 2 ? 2 : 0) : (s <= 1 ? 3 : 0))
                  : (NS == 9 ? (s <= 4 ? 1 : 0) : (s <= 1 ? 2 : (s == 2 ? 1 : 0)));
}
constexpr int c3_halo_first(int s, int NS, int HP) { int n = 0; for (int t = 0; t < s; ++t) n += c3_halo_pieces(t, NS, HP); return n; }
// DMA instructions a wave issues after "its part of slab g+2", as seen at the wait of step g (ordinal s of its unit):
// the halo pieces of step g+2-R plus everything of steps g+3-R .. g-1.  Every step issues IPW slab instructions plus its
// halo pieces.  At ordinal NS-2 the wait must also cover the last halo piece: the following step reads the next halo.
// (NHB 3: the next unit's halo was issued during the PREVIOUS unit, i.e. before every slab this count skips -- no cap.)
constexpr int c3_younger(int R, int IPW, int s, int NS, int HP, int NHB = 2) {
  int n = (R - 3) * IPW;
  for (int d = 1; d <= R - 2; ++d) n += c3_halo_pieces((((s - d) % NS) + NS) % NS, NS, HP);
  const int cap = (NS - 3 - c3_halo_last(NS, HP)) * IPW;
  if (NHB == 2 && s == NS - 2 && n > cap) n = cap;
  return n;
}
// ---- tile bodies with units of DIFFERENT step counts (round 4, 8 x 32 tiles only) --------------------------------------------------
// A tile is normally nchunk units of the same kind (9 or 6 steps).  Three layer classes carry structural zeros in that form:
//   stride (1,2) on the pixel-pair view (rd_api.hip): the even pixel's 32-channel chunks only meet the three taps dw = 0, but ran
//     all six pair-view taps (three of them on zero weights: 25 % of the layer's MFMAs);
//   the 72-channel input of the level-0 tower convs / the 8-channel first layer: a chunk with <= 16 real channels ran two 16-channel
//     k-steps per tap, one of them on zeros.
// A BODY is a fixed cyclic sequence of up to four units of possibly different kinds; the tile runs it a.nchunk / NU times.  Every
// compile-time table of the homogeneous form (tap of a step, halo pieces per step, counted waits) becomes a function of the GLOBAL
// step index g within the body, evaluated cyclically (the step before step 0 is the body's last step: of the previous repetition
// or of the previous tile).
//   UK_T9 / UK_T6A / UK_T6B: the 9-tap and the two 6-tap units of the homogeneous forms (TS 0 / 1 / 2)
//   UK_T3: 3 steps -- the three taps (dh, dw = 0) of the pair view (index 1 of the TS 1 column numbering): an even-pixel chunk
//   UK_P5: 5 steps -- k-step 0 of step s is tap 2s, k-step 1 tap 2s + 1 (tap 9: zero weights), both on the chunk's FIRST 16
//          channels: a chunk with at most 16 real channels in 10 instead of 18 k-steps
// BODY 1 (stride 2), per 128-byte line of the view pixel: [T6A odd chunk | T3 even | T3 even | T6A odd] -- the chunk whose lines
//   are new to L2 is always fetched during a 6-step unit, the 3-step units fetch the second half of a line that is already there;
// BODY 2: [T9, T9, P5] (64 + <= 16 channels: the level-0 tower convs on [agg3 | range image]);  BODY 3: [P5] (the first layer).
constexpr int C3_BODY_M16 = 16;   // launch_conv3's `body` argument: not a tile body but the v_mfma_f32_16x16x32 form of the homogeneous one
enum { UK_T9 = 0, UK_T6A = 1, UK_T6B = 2, UK_T6R = 3, UK_T3 = 4, UK_P5 = 5 };
constexpr int uk_nsteps(int k) { return k == UK_T9 ? 9 : k == UK_T3 ? 3 : k == UK_P5 ? 5 : 6; }
struct C3KStep { int dh, dw, slot; };   // tap row, column index into the per-lane column offsets, 16-channel slot of the chunk
constexpr C3KStep uk_kstep(int k, int s, int ks) {
  if (k == UK_P5) { const int t = 2 * s + ks > 8 ? 8 : 2 * s + ks; return C3KStep{t / 3, t % 3, 0}; }
  const int T = k == UK_T9 ? s : k == UK_T3 ? 3 * s + 1 : 3 * (s / 2) + (s % 2) + (k == UK_T6B ? 1 : 0);
  return C3KStep{T / 3, T % 3, ks};
}
struct C3Body { int nu; int kind[4]; };
constexpr C3Body c3_body(int BODY) {
  return BODY == 1 ? C3Body{4, {UK_T6A, UK_T3, UK_T3, UK_T6A}} : BODY == 2 ? C3Body{3, {UK_T9, UK_T9, UK_P5, 0}} : C3Body{1, {UK_P5, 0, 0, 0}};
}
constexpr int c3b_steps(int BODY) { int n = 0; for (int u = 0; u < c3_body(BODY).nu; ++u) n += uk_nsteps(c3_body(BODY).kind[u]); return n; }
constexpr int c3b_unit(int BODY, int g) { int u = 0; while (g >= uk_nsteps(c3_body(BODY).kind[u])) { g -= uk_nsteps(c3_body(BODY).kind[u]); ++u; } return u; }
constexpr int c3b_ord(int BODY, int g) { int u = 0; while (g >= uk_nsteps(c3_body(BODY).kind[u])) { g -= uk_nsteps(c3_body(BODY).kind[u]); ++u; } return g; }
// halo pieces (wide tile: 6 per wave and unit) issued in step s of a unit with NS steps -- all of them early enough for the wait at
// ordinal NS - 2 to cover them
constexpr int uk_halo_pieces(int NS, int s) {
  return NS == 9 ? (s == 0 ? 2 : s <= 4 ? 1 : 0) : NS == 6 ? (s <= 2 ? 2 : 0) : NS == 5 ? (s <= 1 ? 3 : 0) : (s == 0 ? 6 : 0);
}
constexpr int uk_halo_last(int NS) { return NS == 9 ? 4 : NS == 6 ? 2 : NS == 5 ? 1 : 0; }
constexpr int c3b_pieces(int BODY, int g) { return uk_halo_pieces(uk_nsteps(c3_body(BODY).kind[c3b_unit(BODY, g)]), c3b_ord(BODY, g)); }
constexpr int c3b_first(int BODY, int g) { int n = 0; for (int t = g - c3b_ord(BODY, g); t < g; ++t) n += c3b_pieces(BODY, t); return n; }
// counted wait of global step g (c3_younger above, with the look-back running cyclically over the body)
constexpr int c3b_younger(int BODY, int R, int IPW, int g) {
  const int G = c3b_steps(BODY);
  int n = (R - 3) * IPW;
  for (int d = 1; d <= R - 2; ++d) n += c3b_pieces(BODY, ((g - d) % G + G) % G);
  const int NS = uk_nsteps(c3_body(BODY).kind[c3b_unit(BODY, g)]), s = c3b_ord(BODY, g);
  const int cap = (NS - 3 - uk_halo_last(NS)) * IPW;
  return (s == NS - 2 && n > cap) ? cap : n;
}
// unit i of a tile (i = 0 .. nchunk - 1 in program order) -> 32-channel chunk of the (view) input it reads
__host__ __device__ inline int c3b_chunk(int BODY, int i, int nchunk) {
  if (BODY != 1) return i;
  const int r2 = (i >> 2) * 2, j = i & 3, ne = nchunk >> 1;
  return j == 0 ? ne + r2 : j == 1 ? r2 : j == 2 ? r2 + 1 : ne + r2 + 1;
}
// packed weights of a heterogeneous body: one slab per step in PROGRAM order, slab = [ks (2)][Cout/32][64 lanes][8] like
// pack_taps_frag; k-step ks of step s of a unit reads channels 32*chunk + 16*slot + 8*hi + j of tap (dh, dw) (uk_kstep).
// get(co, ci, dh, dw) -> float with dh, dw in 0..2 (dw: the unit kind's column index).  cin = channels of the (view) input.
template <class F>
inline void pack_body_frag(int BODY, int cin, int cout, void* out, F get, int dt = RD_BF16) {
  const C3Body bd = c3_body(BODY);
  const int nchunk = (cin + 31) / 32, ncb = cout / 32;
  bf16_t* o = (bf16_t*)out;
  for (int i = 0; i < nchunk; ++i) {
    const int kind = bd.kind[i % bd.nu], chunk = c3b_chunk(BODY, i, nchunk);
    for (int s = 0; s < uk_nsteps(kind); ++s)
      for (int ks = 0; ks < 2; ++ks) {
        const C3KStep k = uk_kstep(kind, s, ks);
        const bool zero_tap = kind == UK_P5 && 2 * s + ks > 8;
        for (int cb = 0; cb < ncb; ++cb)
          for (int lane = 0; lane < 64; ++lane) {
            const int co = cb * 32 + conv_row_perm(lane & 31);
            for (int j = 0; j < 8; ++j) {
              const int ci = chunk * 32 + k.slot * 16 + (lane >> 5) * 8 + j;
              *o++ = h16_from_f32(dt, (ci < cin && !zero_tap) ? get(co, ci, k.dh, k.dw) : 0.f);
            }
          }
      }
  }
}
// Which body a folded-scale launch on the 8 x 32 tiles takes (the host packers and the launcher must agree): 1 = stride (1,2) on the
// pixel-pair view whose even / odd halves are whole 128-byte lines (x_cstride a multiple of 64, all of them convolved),
// 2 = [64 channels | <= 16 channels] (rd_conv3x3_bn_act_cat), 3 = at most 16 input channels, 0 = homogeneous units.
// RD_CONV_BODY=0 (dev switch, A/B): always 0.
inline bool conv3_bodies_on() {
  const DevSwitches& sw_ = dev_switches();
  return sw_.conv_body && sw_.conv_wide && sw_.conv_w30 == 2 && sw_.conv_th4 == 1 && !sw_.conv_v1;
}
inline int conv3_body_s2(int cin, int x_cstride, bool folded) { return folded && conv3_bodies_on() && x_cstride % 64 == 0 && cin == x_cstride ? 1 : 0; }
inline int conv3_body_small(int cin, bool folded) { return folded && conv3_bodies_on() && cin <= 16 ? 3 : 0; }
inline int conv3_body_cat(int cin1, int cin2, bool folded) { return folded && conv3_bodies_on() && cin1 == 64 && cin2 <= 16 ? 2 : 0; }
// s_waitcnt immediate (gfx9): vmcnt <= vm and lgkmcnt <= lgkm, expcnt untouched
#define C3_WAIT_IMM(vm, lgkm) (((vm) & 15) | (((vm) >> 4) << 14) | (7 << 4) | ((lgkm) << 8))

// HEAD: the conv is the last layer of a head tower and its only consumer is the tower's 1x1 output conv (<= 8 outputs).
// That conv is applied to each 32-pixel fragment while it sits, already rounded to bf16, in the epilogue's transpose
// scratch (same LDS image and MFMA scheme as head_out_mfma_kernel, k_misc.h: weights as a bf16 hi + lo pair), and the
// 128-channel result is never written to HBM.
// FOLD: the BatchNorm scale is folded into the packed weights (a.scale == nullptr) and the shift enters the accumulators
// through one rank-1 MFMA per accumulator at the start of every tile (A = the shift as a bf16 high + low pair in k = 0, 1;
// B = ones) -- 4 * NCT MFMAs of the tile's 36 * 8 * NCT..., in exchange for which the epilogue has no multiply-add left:
// it reads the accumulators, adds the residual if any, converts and clamps.
// DT = RD_BF16 or RD_F16: the element type of activations and weights (same layouts; the MFMA instruction and the conversions
// of the epilogue differ, rd_common.h H16<DT>).
// GRP: two problems per launch (Conv3Args::ngrp): the tile list runs over 2B images, the group of an image selects the input,
// weight image, shift, residual, output (and fused output conv) -- the launch then has twice the tiles per resident workgroup
// slot (half the tail round) and one prologue / drain instead of two.  Only for the forms the lowering pairs: FOLD, no SC.
// M16 (round 6): the MFMAs as v_mfma_f32_16x16x32 instead of 32x32x16.  Under the part's power cap the matrix cores sustain more of the
// SAME arithmetic in that shape (tools/micro/mfma_power.hip, profiles/r06g_mfma_power.txt: operands from LDS at the same bytes per FLOP,
// post-ReLU activations against N(0, 0.05) weights: 1 539 -> 1 698 TFLOP/s, shader clock 1 753 -> 2 015 MHz; a 16 x 16 tile moves half the
// accumulator bits per FLOP).  Cout 128 on the 8 x 32 tiles only.  A step (one tap of one 32-channel chunk) is ONE k-step of 32 channels:
// 4 pixel half-fragments (16 pixels x 32 channels: lane (n, q) reads slot q of pixel n -- the same LDS bytes as the two 16-channel
// k-steps of a 32-pixel fragment) against the slab's 8 weight fragments (16 channels x 32: pack_taps_frag16) = 32 MFMAs on 32
// accumulators of 4 registers.  Block 0 = weight fragments 0..3, block 1 = 4..7, pixel-major inside a block, so a pixel half-fragment's
// register is free after its 4th MFMA of block 1 and is re-read for the next step right there (single buffered); the barrier, the counted
// waits and the DMA schedule are those of the 32 x 32 form (12 fragment reads per step in both).
template <int NCT, int DBG = 0, int TS = 0, bool HEAD = false, bool SC = false, bool FOLD = false, int FPW = 4, int FC = 2, int NHB = 2,
          int DT = RD_BF16, bool WD = false, bool GRP = false, int BODY = 0, bool M16 = false>
__global__ __launch_bounds__(256, (FPW == 2 ? 2 : 1)) void conv3x3_stream_kernel(Conv3Args a) {
  static_assert(!M16 || (NCT == 4 && FPW == 2 && FC == 1 && WD && FOLD && !SC && !GRP && TS == 0 && BODY == 0 && DBG == 0),
                "16 x 16 x 32 form: cout 128 on the 8 x 32 tiles, folded scales, all nine taps");
  static_assert(BODY == 0 || (WD && FOLD && !GRP && !HEAD && (TS == 0 || TS == 1)), "heterogeneous tile bodies: 8 x 32 tiles, folded scales");
  static_assert(!GRP || (FOLD && !SC && !(HEAD && FPW == 4)), "two problems per launch: folded scales, no shortcut, output-conv weights from L2");
  static_assert((NHB == 2 && !WD) || FOLD, "three halo buffers / wide tile: no room for the scale / shift array");
  static_assert(!HEAD || (NCT == 4 && TS == 0 && (FPW == 4 || FC == 1)), "fused output conv: cout 128, all nine taps, 8-row tiles");
  constexpr bool PH = TS == 3;                   // all phases of a transposed conv: (tile, phase) list, run-time tap-set side
  static_assert(!PH || (FOLD && !SC && !HEAD && !GRP), "all-phase transposed conv: folded scales, no shortcut / output conv / second problem");
  static_assert(!(HEAD && SC), "a head tower has no shortcut");
  static_assert(FPW == 4 || FPW == 2, "4 or 2 pixel fragments per wave");
  static_assert(FC == 2 || (FC == 1 && FPW == 2), "30-column tiles: two fragments per wave");
  using Cfg = C3Cfg<NCT, FPW, FC, NHB, WD>;
  constexpr int HPC = Cfg::HPC;
  constexpr int R = Cfg::R, IPW = Cfg::IPW, SLAB = Cfg::SLAB, COUT = NCT * 32;
  constexpr int C3_HALO = Cfg::HALO, C3_HPW = Cfg::HPW, C3_TH = Cfg::TH;   // (shadow the FPW 4 file-scope constants)
  constexpr int C3_TW = Cfg::TW, C3_ROWB = Cfg::ROWB, RW = Cfg::RW;
  constexpr int NR = FPW + NCT;                  // fragment reads per k-step
  constexpr int NM = FPW * NCT;                  // MFMAs per k-step
  constexpr int NS = c3_nsteps(TS);              // steps (taps) per unit (BODY 0)
  constexpr int BG = BODY ? c3b_steps(BODY) : NS;          // steps of one repetition of the tile body
  constexpr int BNU = BODY ? c3_body(BODY).nu : 1;         // units of one repetition
  HIP_DYNAMIC_SHARED(unsigned char, smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, hi = lane >> 5;
  constexpr int RING = NHB * C3_HALO;
  // byte offset of the halo buffer after / before buffer a in the cycle
  auto hnext = [](int a_) { return NHB == 2 ? C3_HALO - a_ : (a_ == (NHB - 1) * C3_HALO ? 0 : a_ + C3_HALO); };
  auto hprev = [](int a_) { return NHB == 2 ? C3_HALO - a_ : (a_ == 0 ? (NHB - 1) * C3_HALO : a_ - C3_HALO); };
  float* Sc = (float*)(smem + RING + R * SLAB);
  constexpr int HWOFF = RING + R * SLAB + 2 * COUT * (int)sizeof(float);   // HEAD: 16 KB of packed output-conv weights
  if (NHB == 2 && !WD && tid < COUT) {
    Sc[tid] = a.scale ? a.scale[tid] : 1.f;
    Sc[COUT + tid] = a.shift ? a.shift[tid] : 0.f;
  }
  // FOLD: per channel block j the dword {bf16 hi, bf16 lo} of this lane's shift (A-operand row m = channel
  // 32*j + conv_row_perm(m), k = 0 and 1 live in the hi == 0 half of the wave), and the B operand of ones
  unsigned bzw[NCT];
  auto load_shift = [&](const float* shp) {
#pragma unroll
    for (int j = 0; j < NCT; ++j) {
      const float t = shp ? shp[32 * j + conv_row_perm(m)] : 0.f;
      const bf16_t th = H16<DT>::from_f32(t);
      const bf16_t tl = H16<DT>::from_f32(t - H16<DT>::to_f32(th));
      bzw[j] = hi ? 0u : ((unsigned)th | ((unsigned)tl << 16));
    }
  };
  if constexpr (FOLD && !GRP && !M16) load_shift(a.shift);
  // GRP: the fragment of BOTH problems, loaded once (4 more registers; a reload inside the tile loop would be a vector-memory
  // load whose wait drains the LDS-DMA queue -- measured +8 .. 12 us per launch); the tile's one is selected per tile
  unsigned bzw1[GRP ? NCT : 1];
  if constexpr (GRP) {
    load_shift(a.shift ? a.shift + a.g_shift : nullptr);
#pragma unroll
    for (int j = 0; j < NCT; ++j) bzw1[j] = bzw[j];
    load_shift(a.shift);
  }
  // M16: the shift of MFMA row lane & 15 of fragment cb as the dword {hi, lo}, held by lane quad cb & 3 in register cb >> 2: its two
  // halves are k = 8*(cb & 3), +1 of the A operand, and the rank-1 MFMA of fragment cb takes ones in exactly those two k (2 registers
  // instead of 8 across the MFMA phase)
  unsigned bzw16[M16 ? 2 : 1];
  if constexpr (M16) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float t = a.shift ? a.shift[conv_row16(4 * g + (lane >> 4), lane & 15)] : 0.f;
      const bf16_t th = H16<DT>::from_f32(t);
      const bf16_t tl = H16<DT>::from_f32(t - H16<DT>::to_f32(th));
      bzw16[g] = (unsigned)th | ((unsigned)tl << 16);
    }
  }
  int tpt = 0;
#define C3_TRACE() { if (a.trace && tid == 0 && tpt < 7) a.trace[(size_t)blockIdx.x * 8 + tpt++] = wall_clock64(); }
  C3_TRACE()
  const unsigned long long clk0 = __builtin_readcyclecounter();

  const int G = gridDim.x, wg = blockIdx.x;
  const int nph = PH ? a.nph : 1;
  const int ntl = ((a.ntiles - wg + G - 1) / G) * nph;   // list entries of this workgroup: spatial tiles wg, wg + G, ... (grid <= ntiles), PH: x phases
  const int tiles_img = a.ncol * a.nrow;

  // ---- DMA issue -----------------------------------------------------------------------------------------------
  // LDS-DMA with a precomputed LDS address in M0 (nothing else in this kernel uses M0).  Issued through inline asm so
  // that hipcc's waitcnt pass does not see it (k_conv.h / rd_common.h lds_dma16 explain why); hipemu uses the builtin.
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem);
#endif
  auto dma_s = [&](const unsigned char* sbase, unsigned voff, int lds_off) {   // uniform base + 32-bit lane offset
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds0 + (unsigned)lds_off)) : "memory");   // (uniform by construction; the intrinsic keeps it in an SGPR whatever the allocator made of the buffer cursor)
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
  // halo image: linear 16-B slot P = 4*px + ps of the LDS image holds logical slot s = ps ^ ((px >> 2) & 3) of halo
  // pixel px = 64*r + cc (conflict-free ds_read_b128 for any tap shift).  Wave w issues the 10
  // 1-KB pieces q = 10*w .. 10*w+9: px = 16*q + (lane >> 2), so r = q >> 2 and cc = 16*(q & 3) + (lane >> 2).
  // Source of a piece = uniform address of halo pixel (r, 16*(q&3)) + a per-lane constant; lanes outside the image
  // (or past the last channel slot) read the zero page instead.
  const int hs = (lane & 3) ^ ((lane >> 4) & 3);          // logical slot this lane fetches (same for every piece)
  const int l4 = lane >> 2;
  const unsigned hlane = (unsigned)(l4 * a.x_cs + hs * 8) * 2;
  int hk = 0, hc = 0;                                     // (tile ordinal, chunk) of the NEXT halo to fetch
  int hh0 = 0, hlo = 0, hlim = 0;                         // first halo row; valid halo columns are hlo <= cc < hlim
  const unsigned char* hbase = nullptr;                   // uniform: halo pixel (0, 0), channel slot 4*hc
  bool hsok = false;
  int hcur = 0;                                           // chunk of the unit being fetched (WD: the slot test is per piece)
  // Tile t = wg + k*G of the list -> (column tile, row block, image).  Decoded ONCE for the workgroup's first tile; every later
  // tile is the previous one plus G in mixed radix (ncol, nrow) -- a few scalar adds and selects instead of the three software
  // integer divisions (~60 dependent scalar instructions in front of a wave's MFMAs, once per unit) a decode from t costs.
  // (XCD-aware order: j advances by G/8 per list step -> row block by (G/8) % nrow, strip by 8*dq or 8*(dq + 1))
  const int xdq = (G >> 3) / a.nrow;
  const int g_ct = a.xcd ? (8 * xdq) % a.ncol : G % a.ncol, g_rb = a.xcd ? (G >> 3) % a.nrow : (G / a.ncol) % a.nrow,
            g_b = a.xcd ? (8 * xdq) / a.ncol : G / tiles_img;
  const int g_ct1 = (8 * xdq + 8) % a.ncol, g_b1 = (8 * xdq + 8) / a.ncol;          // strip step with the row-block carry
  auto tile_advance = [&](int& ct, int& rb, int& b) __attribute__((always_inline)) {
    // one select-only form for both orders: digit 0 (radix r0) carries into digit 1 (radix r1) carries into the image index
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
  const int xs0 = ((wg >> 3) / a.nrow) * 8 + (wg & 7);                             // (XCD-aware order) first strip of this workgroup
  int f_ct = a.xcd ? xs0 % a.ncol : wg % a.ncol, f_rb = a.xcd ? (wg >> 3) % a.nrow : (wg / a.ncol) % a.nrow,
      f_b = a.xcd ? xs0 / a.ncol : wg / tiles_img;                                  // tile of the NEXT halo to fetch
  int c_ct = f_ct, c_rb = f_rb, c_b = f_b;                                        // tile being computed
  int f_ph = 0, c_ph = 0, s_ph = 0;                                               // (PH) phase of the fetch / compute / slab cursor
  const unsigned char* htile = nullptr;                   // uniform: halo pixel (0, 0) of the fetch tile, channel 0
  const unsigned char* htile2 = nullptr;                  // (two-tensor input) the same pixel of x2
  int hpxb = a.x_cs * 2, hns = 0;                         // bytes per pixel and valid 16-byte slots of the chunk being fetched
  auto halo_tile = [&]() {                                // per fetch TILE: geometry and base address
    const int hw0 = f_ct * C3_TW - 1;
    hh0 = f_rb * C3_TH - 1;
    hlo = hw0 < 0 ? -hw0 : 0;
    hlim = a.W - hw0;
    const bool fg = GRP && f_b >= a.B;                     // (GRP) problem 1: its own input, image index from 0
    htile = (const unsigned char*)(a.x + (size_t)(fg ? f_b - a.B : f_b) * a.x_bs + a.x_co + (fg ? a.g_x : 0)) +
            ((long)hh0 * a.W + hw0) * (long)a.x_cs * 2;
    if constexpr (WD) {
      if (a.x2) htile2 = (const unsigned char*)(a.x2 + (size_t)f_b * a.x2_bs + a.x2_co) + ((long)hh0 * a.W + hw0) * (long)a.x2_cs * 2;
    }
  };
  bool hnew = true;                                       // the fetch cursor moved to a new tile: geometry not yet derived
  auto halo_begin = [&]() {                               // per fetch UNIT: the chunk's slice of the tile; advance the cursor
    if (hnew) { halo_tile(); hnew = false; }              // (the pieces of this unit are issued AFTER this call and use the geometry)
    // BODY 1: unit hc of the tile -> 32-channel chunk of the view pixel ([odd | even | even | odd] per 128-byte line, see c3_body)
    const int hcm = c3b_chunk(BODY, hc, a.nchunk);
    hbase = htile + hcm * 64;
    hsok = hcm * 4 + hs < a.nslots;
    hcur = hcm;
    if constexpr (WD) {
      hpxb = a.x_cs * 2; hns = a.nslots - 4 * hcm;
      if (a.x2 && hcm >= a.nchunk1) { hbase = htile2 + (hcm - a.nchunk1) * 64; hpxb = a.x2_cs * 2; hns = a.nslots2 - 4 * (hcm - a.nchunk1); }
    }
    // past the end of the list re-fetch the last unit (keeps the DMA count per step constant)
    if (hc + 1 < a.nchunk) ++hc;
    else if (hk + 1 < ntl) {
      ++hk; hc = 0;
      if constexpr (PH) {   // next phase of the same tile (same geometry, same source lines: L2 hits), then the next tile
        if (++f_ph == nph) { f_ph = 0; tile_advance(f_ct, f_rb, f_b); hnew = true; }
      } else { tile_advance(f_ct, f_rb, f_b); hnew = true; }
    }
  };
  auto halo_piece = [&](int buf, int j) {
    const int q = wave * C3_HPW + j;
    bool ok;
    const unsigned char* src;
    if constexpr (WD) {
      // row pitch 34 pixels: this lane's halo pixel p = 16 q + (lane >> 2) = (r, c) with r = p / 34 (exact for p < 400 as
      // (p * 1928) >> 16); the slot swizzle is by COLUMN, ((c >> 2) & 3), so that a fragment read sees the same bank pattern in
      // every row; pixels past the image's 340 (the padding pieces) read the zero page
      // (through an opaque copy of the lane id: per piece index everything below is loop invariant, and hipcc would keep the six
      //  pieces' rows / columns / offsets live across the MFMA phase -- 30+ spilled registers on the cout-128 form -- instead of
      //  recomputing a dozen VALU instructions per piece)
      int ol = lane;
      asm volatile("" : "+v"(ol));
      const int pp = 16 * q + (ol >> 2), r = (pp * 1928) >> 16, cc = pp - 34 * r;
      const int hsp = (ol & 3) ^ ((cc >> 2) & 3);
      ok = hsp < hns && (unsigned)(hh0 + r) < (unsigned)a.H && cc >= hlo && cc < hlim && pp < Cfg::NPX && !(DBG & 16);
      src = hbase + (long)((r * a.W + cc) * hpxb + hsp * 16);
    } else {
    const int r = q / (2 * FC), c16 = (q % (2 * FC)) * 16;
    const int cc = c16 + l4;
    ok = hsok && (unsigned)(hh0 + r) < (unsigned)a.H && cc >= hlo && cc < hlim && !(DBG & 16);
    const unsigned char* sp = hbase + ((long)r * a.W + c16) * (long)a.x_cs * 2;
    src = sp + hlane;
    }
    if constexpr ((DBG & 256) != 0)   // ablation: the same lanes' data from the first MB of x (L2-resident REAL data: MFMA power as in production, no HBM reads)
      src = (const unsigned char*)a.x + ((size_t)(src - (const unsigned char*)a.x) & 0xFFFF0);
    dma_v(ok ? (const void*)src : (const void*)a.zero16, buf + q * 1024);   // (buf = byte offset of the halo buffer)
  };
  int fslot = 0, fslab = 0;                               // ring slot / slab-within-tile of the NEXT slab to fetch
  const int nslab_tile = BODY ? (a.nchunk / BNU) * BG : a.nchunk * NS;
  // GRP: the slab stream runs R steps ahead of the MFMAs, so it has its own tile cursor -- the weight image is the one of the
  // problem that tile belongs to (past the end of the list the dummy fetches read whichever image the cursor has reached)
  // (n0 = how many of this workgroup's tiles wg, wg + G, ... belong to problem 0, i.e. lie below tiles_img * B: one division per
  //  workgroup instead of a third mixed-radix tile cursor advanced in every step)
  const int t0_ = tiles_img * a.B;
  const int n0 = GRP ? (wg < t0_ ? (t0_ - wg + G - 1) / G : 0) : 0;
  int stile = 0;                                          // ordinal (within the workgroup's list) of the tile the slab stream is in
  const unsigned char* wsrc = a.w + (GRP && n0 == 0 ? a.g_w : 0);
  auto slab_piece = [&](int j) {
    dma_s(wsrc + (size_t)fslab * SLAB + (wave * IPW + j) * 1024, lane * 16, RING + fslot * SLAB + (wave * IPW + j) * 1024);
  };
  auto slab_advance = [&]() {
    fslot = fslot + 1 == R ? 0 : fslot + 1;
    fslab = fslab + 1 == nslab_tile ? 0 : fslab + 1;
    if constexpr (GRP) {
      if (fslab == 0) {
        ++stile;
        wsrc = a.w + (stile >= n0 ? a.g_w : 0);
      }
    }
    if constexpr (PH) {   // the slab stream's own phase cursor (it runs R steps ahead of the MFMAs)
      if (fslab == 0) {
        s_ph = s_ph + 1 == nph ? 0 : s_ph + 1;
        wsrc = a.w + (size_t)s_ph * a.w_pb;
      }
    }
  };

  // ---- fragment addressing -------------------------------------------------------------------------------------
  // pixel fragment i of tap (dh, dw): halo pixel (2*wave + (i >> 1) + 1 + dh, 1 + dw + 32*(i & 1) + m), px = 64*row + col
  // (so +32 px = +2048 B per fragment, +4096 B per tap row, +8192 B per wave); byte = px*64 + ((slot ^ (px>>2)) & 3)*16
  // with slot = 2*ks + hi.  (px >> 2) & 3 only depends on (1 + dw + m), +32 px = +2048 B, ks toggles bit 5.
  int aoff[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int c = d + m;                                  // 1 + dw + m with dw = d - 1
    aoff[d] = c * 64 + (((hi ^ (c >> 2)) & 3) << 4) + wave * RW * C3_ROWB;
    if constexpr (M16) {                                  // lane (n, q): slot q of pixel column d + n (+16 columns = +1024 B, same swizzle term)
      const int c16 = d + (lane & 15);
      aoff[d] = c16 * 64 + ((((lane >> 4) ^ (c16 >> 2)) & 3) << 4) + wave * RW * C3_ROWB;
    }
  }
  const int boff = RING + lane * 16;
  // PH: the two column offsets of the tile's tap-set side; pq0 = column 0 of the NEXT list entry's side (the last step of a tile
  // pre-reads the first fragment of the next one)
  auto side2 = [&](int ph) { return (a.ts_mask >> ph) & 1; };
  int pcol[2] = {aoff[0], aoff[1]}, pq0 = aoff[0];
  if constexpr (PH) {
    const int s0_ = side2(0), s1_ = side2(nph > 1 ? 1 : 0);
    pcol[0] = s0_ ? aoff[1] : aoff[0]; pcol[1] = s0_ ? aoff[2] : aoff[1];
    pq0 = s1_ ? aoff[1] : aoff[0];
  }
  bool lastc = false;   // (PH) the unit being computed is the last chunk of its tile
#define C3_AO(D) (PH ? pcol[(D)] : aoff[(D)])

  f32x16 acc[M16 ? 1 : FPW][M16 ? 1 : NCT];
  f32x4 acc6[M16 ? 4 : 1][M16 ? 8 : 1];   // M16: [pixel half-fragment 2*row + half][16-channel fragment]
  s16x8 fa[2][FPW], fb[2][NCT];
  s16x8 fa16[4];
#define C3_FENCE() __builtin_amdgcn_sched_barrier(0)
  // fragment read k of a k-step, in the order the MFMA sequence needs them: fa[0], fb[0..NCT-1], fa[1..3]
#define C3_RD(BUF, K, AADDR, BADDR, KS)                                                                   \
  {                                                                                                       \
    if ((K) == 0) fa[BUF][0] = *(const s16x8*)(smem + (AADDR));                                            \
    else if ((K) <= NCT) fb[BUF][(K) - 1] = *(const s16x8*)(smem + (BADDR) + ((KS) * NCT + (K) - 1) * 1024); \
    else fa[BUF][(K) - NCT] = *(const s16x8*)(smem + (AADDR) + (((K) - NCT) / FC) * C3_ROWB + (((K) - NCT) % FC) * 2048); \
    C3_FENCE();                                                                                           \
  }
  // MFMA n of a k-step: pixel fragment n / NCT against channel fragment n % NCT (transposed: weights are operand A)
#define C3_MM(BUF, N)                                                                                     \
  {                                                                                                       \
    if (!(DBG & 8))                                                                                       \
      acc[(N) / NCT][(N) % NCT] = H16<DT>::mfma(fb[BUF][(N) % NCT], fa[BUF][(N) / NCT], \
                                                                          acc[(N) / NCT][(N) % NCT]);    \
    C3_FENCE();                                                                                           \
  }
  // first k-step of a tile: C = 0 as an inline constant instead of zeroing 16 * 4 * NCT accumulator registers per tile
#define C3_MMZ(BUF, N)                                                                                    \
  {                                                                                                       \
    if (!(DBG & 8))                                                                                       \
      acc[(N) / NCT][(N) % NCT] = H16<DT>::mfma(fb[BUF][(N) % NCT], fa[BUF][(N) / NCT], \
                                                                          f32x16{});             \
    C3_FENCE();                                                                                           \
  }
  // M16: weight fragment C of the slab half at BADDR; pixel half-fragment HX = 2*row + half of the tap at AADDR; MFMA N of block BK =
  // pixel half-fragment N / 4 against weight fragment 4*BK + N % 4
#define C3_RDW16(BUF, C, BADDR) { fb[BUF][C] = *(const s16x8*)(smem + (BADDR) + (C) * 1024); C3_FENCE(); }
#define C3_RDP16(HX, AADDR) { fa16[HX] = *(const s16x8*)(smem + (AADDR) + ((HX) >> 1) * C3_ROWB + ((HX) & 1) * 1024); C3_FENCE(); }
#define C3_MM16(BK, N)                                                                                                   \
  {                                                                                                                      \
    acc6[(N) >> 2][4 * (BK) + ((N) & 3)] = H16<DT>::mfma16(fb[BK][(N) & 3], fa16[(N) >> 2], acc6[(N) >> 2][4 * (BK) + ((N) & 3)]); \
    C3_FENCE();                                                                                                          \
  }
  // workgroup barrier that retires the counted DMA and all LDS reads except the NR youngest (the fragments of the next
  // step just requested), nothing else (see k_conv.h for why this is not __syncthreads())
#define C3_SYNC(VMCNT, LGKM)                                      \
  {                                                               \
    asm volatile("" ::: "memory");                                \
    __builtin_amdgcn_s_waitcnt(C3_WAIT_IMM((DBG & 32) ? 63 : (VMCNT), LGKM)); \
    if (!(DBG & 2)) __builtin_amdgcn_s_barrier();                 \
    asm volatile("" ::: "memory");                                \
    C3_FENCE();                                                   \
  }

  // ---- prologue: first halo, a full ring ------------------------------------------------------------------------
  halo_begin();
#pragma unroll
  for (int j = 0; j < C3_HPW; ++j) halo_piece(0, j);
  if constexpr (NHB == 3) {   // the fetch runs two units ahead: unit 1 as well
    halo_begin();
#pragma unroll
    for (int j = 0; j < C3_HPW; ++j) halo_piece(C3_HALO, j);
  }
  if constexpr (HEAD && FPW == 4) {   // (FPW 2: no LDS to spare for them with two workgroups per CU -- read from L2 in the epilogue)
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_s(a.hw + (wave * 4 + j) * 1024, lane * 16, HWOFF + (wave * 4 + j) * 1024);
  }
#pragma unroll 1
  for (int s0 = 0; s0 < R; ++s0) {
#pragma unroll
    for (int j = 0; j < IPW; ++j) slab_piece(j);
    slab_advance();
  }
  C3_SYNC(0, 0)
  C3_TRACE()
  // tap of the very first k-step (compile-time constants: a run-time evaluation of the body tables would put them on the stack)
  constexpr C3KStep K00 = BODY ? uk_kstep(c3_body(BODY ? BODY : 3).kind[0], 0, 0) : C3KStep{c3_tap(TS, 0) / 3, c3_tap(TS, 0) % 3, 0};
  constexpr int K00DW = K00.dw, K00DH = K00.dh;
  int rslot = 0;        // ring slot of the slab being consumed
  int abuf = 0;         // halo buffer (byte offset) of the unit being consumed
  if constexpr (M16) {
#pragma unroll
    for (int k = 0; k < 4; ++k) C3_RDW16(0, k, boff + rslot * SLAB)
#pragma unroll
    for (int k = 0; k < 4; ++k) C3_RDP16(k, aoff[K00DW] + abuf + K00DH * C3_ROWB)
  } else {
#pragma unroll
  for (int k = 0; k < NR; ++k)   // fragments of (unit 0, first tap, ks 0)
    C3_RD(0, k, C3_AO(K00DW) + abuf + K00DH * C3_ROWB, boff + rslot * SLAB, 0)
  }

  // One step = tap T of the current unit, software pipelined by hand (one wave per SIMD: nothing else hides latency).
  //   block 0: MFMAs of ks 0, with the reads of (this step, ks 1) interleaved 1:1 into its first half;
  //   block 1: MFMAs of ks 1, with the reads of (next step, ks 0) interleaved into its first half, then the step's
  //            barrier (slab g+2 landed everywhere, slab g free), then the DMA issue interleaved into the second half.
#define C3_STEP(S) C3_STEP_(S, C3_MM)
#define C3_STEP_(S, MM0)                                                                                             \
  {                                                                                                                  \
    constexpr int T_ = c3_tap(TS, (S)), TN_ = c3_tap(TS, ((S) + 1) % NS);                                            \
    constexpr int dh_ = T_ / 3, dw_ = T_ % 3, ndh_ = TN_ / 3, ndw_ = TN_ % 3;                                        \
    constexpr int NH_ = c3_halo_pieces((S), NS, HPC), NP_ = IPW + NH_;   /* DMA pieces of this step */                    \
    const int acur_ = (C3_AO(dw_) + abuf + dh_ * C3_ROWB) ^ 32;                                                        \
    const int bcur_ = boff + rslot * SLAB;                                                                           \
    const int rnext_ = rslot + 1 == R ? 0 : rslot + 1;                                                               \
    const int anext_ = ((S) == NS - 1 && PH && lastc ? pq0 : C3_AO(ndw_)) + ((S) == NS - 1 ? hnext(abuf) : abuf) + ndh_ * C3_ROWB; \
    const int bnext_ = boff + rnext_ * SLAB;                                                                         \
    const int hbuf_ = hprev(abuf);   /* NHB 2: the other buffer (unit u+1); NHB 3: unit u-1's, for unit u+2 */       \
    C3_FENCE();                                                                                                      \
    _Pragma("unroll") for (int n = 0; n < NM; ++n) {                                                                 \
      MM0(0, n)                                                                                                      \
      if (n < NR) C3_RD(1, n, acur_, bcur_, 1)                                                                       \
    }                                                                                                                \
    _Pragma("unroll") for (int n = 0; n < NM / 2; ++n) {                                                             \
      C3_MM(1, n)                                                                                                    \
      if (n < NR) C3_RD(0, n, anext_, bnext_, 0)                                                                     \
    }                                                                                                                \
    if (NR > NM / 2) { _Pragma("unroll") for (int n = NM / 2; n < NR; ++n) C3_RD(0, n, anext_, bnext_, 0) }          \
    C3_SYNC(c3_younger(R, IPW, (S), NS, HPC, NHB), NR)                                                            \
    if ((S) == 0) halo_begin();                                                                                      \
    _Pragma("unroll") for (int n = NM / 2; n < NM; ++n) {                                                            \
      C3_MM(1, n)                                                                                                    \
      if (!(DBG & 4)) {   /* the step's DMA pieces, spread over the MFMA slots of this half block (slab pieces first) */ \
        _Pragma("unroll") for (int p = 0; p < NP_; ++p)                                                              \
          if (p * (NM / 2) / NP_ == n - NM / 2) {                                                                    \
            if (p < IPW) slab_piece(p); else halo_piece(hbuf_, c3_halo_first((S), NS, HPC) + p - IPW);                    \
            C3_FENCE();                                                                                              \
          }                                                                                                          \
      }                                                                                                              \
    }                                                                                                                \
    slab_advance();                                                                                                  \
    rslot = rnext_;                                                                                                  \
  }

  // M16 step (see the kernel's header): block 0 with the reads of the slab's second half, block 1 with the reads of the next step
  // (weights first, a pixel half-fragment right after its last MFMA), the barrier after the first half of block 1 (6 reads of the next
  // step outstanding), the DMA issue in its second half.
#define C3_STEP16(S)                                                                                                 \
  {                                                                                                                  \
    constexpr int TN_ = c3_tap(TS, ((S) + 1) % NS), ndh_ = TN_ / 3, ndw_ = TN_ % 3;                                  \
    constexpr int NH_ = c3_halo_pieces((S), NS, HPC), NP_ = IPW + NH_;                                               \
    const int bcur_ = boff + rslot * SLAB;                                                                           \
    const int rnext_ = rslot + 1 == R ? 0 : rslot + 1;                                                               \
    const int anext_ = aoff[ndw_] + ((S) == NS - 1 ? hnext(abuf) : abuf) + ndh_ * C3_ROWB;                           \
    const int bnext_ = boff + rnext_ * SLAB;                                                                         \
    const int hbuf_ = hprev(abuf);                                                                                   \
    C3_FENCE();                                                                                                      \
    _Pragma("unroll") for (int n = 0; n < 16; ++n) {                                                                 \
      C3_MM16(0, n)                                                                                                  \
      if (n < 4) C3_RDW16(1, n, bcur_ + 4096)                                                                        \
    }                                                                                                                \
    _Pragma("unroll") for (int n = 0; n < 8; ++n) {                                                                  \
      C3_MM16(1, n)                                                                                                  \
      if (n < 4) C3_RDW16(0, n, bnext_)                                                                              \
      if (n == 3) C3_RDP16(0, anext_)                                                                                \
      if (n == 7) C3_RDP16(1, anext_)                                                                                \
    }                                                                                                                \
    C3_SYNC(c3_younger(R, IPW, (S), NS, HPC, NHB), 6)                                                                \
    if ((S) == 0) halo_begin();                                                                                      \
    _Pragma("unroll") for (int n = 8; n < 16; ++n) {                                                                 \
      C3_MM16(1, n)                                                                                                  \
      _Pragma("unroll") for (int p = 0; p < NP_; ++p)                                                                \
        if (p * 8 / NP_ == n - 8) {                                                                                  \
          if (p < IPW) slab_piece(p); else halo_piece(hbuf_, c3_halo_first((S), NS, HPC) + p - IPW);                 \
          C3_FENCE();                                                                                                \
        }                                                                                                            \
      if (n == 11) C3_RDP16(2, anext_)                                                                               \
      if (n == 15) C3_RDP16(3, anext_)                                                                               \
    }                                                                                                                \
    slab_advance();                                                                                                  \
    rslot = rnext_;                                                                                                  \
  }

  // Global step g of a heterogeneous body: the same software pipeline as C3_STEP, with the tap / slot of each k-step, the halo
  // pieces and the counted wait taken from the body's tables, and the halo buffer switched after a unit's last step.
#define C3_GSTEP(G_, MM0)                                                                                            \
  if constexpr (BODY != 0 && (G_) < BG) {                                                                            \
    constexpr int U_ = c3b_unit(BODY, (G_)), S_ = c3b_ord(BODY, (G_)), K_ = c3_body(BODY).kind[U_], NSU_ = uk_nsteps(K_); \
    constexpr int GN_ = ((G_) + 1) % BG, KN_ = c3_body(BODY).kind[c3b_unit(BODY, GN_)];                              \
    constexpr C3KStep k1_ = uk_kstep(K_, S_, 1), kn_ = uk_kstep(KN_, c3b_ord(BODY, GN_), 0);                         \
    constexpr bool lastS_ = S_ == NSU_ - 1;                                                                          \
    constexpr int NH_ = c3b_pieces(BODY, (G_)), NP_ = IPW + NH_, HF_ = c3b_first(BODY, (G_)), YG_ = c3b_younger(BODY, R, IPW, (G_)); \
    const int acur_ = (aoff[k1_.dw] + abuf + k1_.dh * C3_ROWB) ^ (k1_.slot << 5);                                    \
    const int bcur_ = boff + rslot * SLAB;                                                                           \
    const int rnext_ = rslot + 1 == R ? 0 : rslot + 1;                                                               \
    const int anext_ = (aoff[kn_.dw] + (lastS_ ? hnext(abuf) : abuf) + kn_.dh * C3_ROWB) ^ (kn_.slot << 5);          \
    const int bnext_ = boff + rnext_ * SLAB;                                                                         \
    const int hbuf_ = hprev(abuf);                                                                                   \
    C3_FENCE();                                                                                                      \
    _Pragma("unroll") for (int n = 0; n < NM; ++n) {                                                                 \
      MM0(0, n)                                                                                                      \
      if (n < NR) C3_RDX(1, n, acur_, bcur_, 1)                                                                      \
    }                                                                                                                \
    _Pragma("unroll") for (int n = 0; n < NM / 2; ++n) {                                                             \
      C3_MM(1, n)                                                                                                    \
      if (n < NR) C3_RDX(0, n, anext_, bnext_, 0)                                                                    \
    }                                                                                                                \
    if (NR > NM / 2) { _Pragma("unroll") for (int n = NM / 2; n < NR; ++n) C3_RDX(0, n, anext_, bnext_, 0) }         \
    C3_SYNC(YG_, NR)                                                                                                 \
    if (S_ == 0) halo_begin();                                                                                       \
    _Pragma("unroll") for (int n = NM / 2; n < NM; ++n) {                                                            \
      C3_MM(1, n)                                                                                                    \
      _Pragma("unroll") for (int p = 0; p < NP_; ++p)                                                                \
        if (p * (NM / 2) / NP_ == n - NM / 2) {                                                                      \
          if (p < IPW) slab_piece(p); else halo_piece(hbuf_, HF_ + p - IPW);                                         \
          C3_FENCE();                                                                                                \
        }                                                                                                            \
    }                                                                                                                \
    slab_advance();                                                                                                  \
    rslot = rnext_;                                                                                                  \
    if (lastS_) abuf = hnext(abuf);                                                                                  \
  }
  // (fragment read with an address that already carries the k-step's slot bit: C3_RD adds nothing for the pixel operand)
#define C3_RDX(BUF, K, AADDR, BADDR, KS) C3_RD(BUF, K, AADDR, BADDR, KS)
#define C3_GBODY_FROM1(MM) C3_GSTEP(1, MM) C3_GSTEP(2, MM) C3_GSTEP(3, MM) C3_GSTEP(4, MM) C3_GSTEP(5, MM) C3_GSTEP(6, MM) C3_GSTEP(7, MM) \
  C3_GSTEP(8, MM) C3_GSTEP(9, MM) C3_GSTEP(10, MM) C3_GSTEP(11, MM) C3_GSTEP(12, MM) C3_GSTEP(13, MM) C3_GSTEP(14, MM) C3_GSTEP(15, MM)      \
  C3_GSTEP(16, MM) C3_GSTEP(17, MM) C3_GSTEP(18, MM) C3_GSTEP(19, MM) C3_GSTEP(20, MM) C3_GSTEP(21, MM) C3_GSTEP(22, MM)
  static_assert(BG <= 23, "C3_GBODY_FROM1 unrolls 23 steps");

  for (int k = 0; k < ntl; ++k) {
    if constexpr (PH) lastc = a.nchunk == 1;
    if constexpr (M16) {
      // every accumulator starts from its channel's shift: one rank-1 MFMA per 16-channel fragment (A = the shift as hi + lo in the two
      // k of lane quad cb & 3, B = ones in the same two k), whose result is the same for every pixel -- the other three pixel
      // half-fragments are register copies
      unsigned z0 = 0u;
      asm volatile("" : "+v"(z0));
      int oq = lane >> 4;
      asm volatile("" : "+v"(oq));
      // (the two shift words through opaque copies as well: as loop invariants hipcc keeps an operand TUPLE across the MFMA phase, spills it
      //  and reloads it here -- a vector-memory load whose wait drains the DMA queue at every tile start; seen in the HEAD form)
      unsigned bq2[2] = {bzw16[0], bzw16[1]};
      asm volatile("" : "+v"(bq2[0]), "+v"(bq2[1]));
#pragma unroll
      for (int cb = 0; cb < 8; ++cb) {
        unsigned ob[4] = {oq == (cb & 3) ? H16<DT>::ONE * 0x10001u : z0, z0, z0, z0};
        unsigned ab[4] = {bq2[cb >> 2], z0, z0, z0};
        s16x8 ones, bz;
        memcpy(&ones, ob, 16);
        memcpy(&bz, ab, 16);
        acc6[0][cb] = H16<DT>::mfma16(bz, ones, f32x4{});
        C3_FENCE();
      }
#pragma unroll
      for (int h = 1; h < 4; ++h)
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) acc6[h][cb] = acc6[0][cb];
      C3_FENCE();
#pragma unroll 1
      for (int c = 0; c < a.nchunk; ++c) {
        C3_STEP16(0) C3_STEP16(1) C3_STEP16(2) C3_STEP16(3) C3_STEP16(4) C3_STEP16(5) C3_STEP16(6) C3_STEP16(7) C3_STEP16(8)
        abuf = hnext(abuf);
      }
    } else {
    {   // first chunk of the tile, peeled: its first k-step starts the accumulators from C = 0 (FOLD: from the shift)
      if constexpr (FOLD) {
        // (the zero words go through an opaque copy once per tile: as compile-time zeros the five operand tuples are loop
        //  invariant, hipcc keeps all 20 registers live across the MFMA phase, spills them, and reloads them here from scratch --
        //  a vector-memory load whose s_waitcnt vmcnt(0) drains the whole DMA queue at every tile start)
        unsigned z0 = 0u;
        asm volatile("" : "+v"(z0));
        const unsigned one2 = hi ? z0 : H16<DT>::ONE * 0x10001u;   // B: k = 0, 1 -> 1.0, the rest 0
        unsigned ob[4] = {one2, z0, z0, z0};
        s16x8 ones;
        memcpy(&ones, ob, 16);
#pragma unroll
        for (int n = 0; n < NM; ++n) {
          unsigned ab[4] = {GRP && k >= n0 ? bzw1[GRP ? n % NCT : 0] : bzw[n % NCT], z0, z0, z0};
          s16x8 bz;
          memcpy(&bz, ab, 16);
          acc[n / NCT][n % NCT] = H16<DT>::mfma(bz, ones, f32x16{});
        }
        C3_FENCE();
        if constexpr (BODY != 0) { C3_GSTEP(0, C3_MM) } else {
        C3_STEP(0)
        }
      } else {
        C3_STEP_(0, C3_MMZ)
      }
      if constexpr (BODY != 0) { C3_GBODY_FROM1(C3_MM) } else {
      C3_STEP(1) C3_STEP(2) C3_STEP(3) C3_STEP(4) C3_STEP(5)
      if constexpr (NS == 9) { C3_STEP(6) C3_STEP(7) C3_STEP(8) }
      abuf = hnext(abuf);
      }
    }
    if constexpr (BODY != 0) {
#pragma unroll 1
      for (int c = BNU; c < a.nchunk; c += BNU) { C3_GSTEP(0, C3_MM) C3_GBODY_FROM1(C3_MM) }
    } else {
#pragma unroll 1
    for (int c = 1; c < a.nchunk; ++c) {
      if constexpr (PH) lastc = c == a.nchunk - 1;
      C3_STEP(0) C3_STEP(1) C3_STEP(2) C3_STEP(3) C3_STEP(4) C3_STEP(5)
      if constexpr (NS == 9) { C3_STEP(6) C3_STEP(7) C3_STEP(8) }
      abuf = hnext(abuf);
    }
    }
    }   // !M16
    if (k == 0) C3_TRACE()

    // ---- epilogue of tile k: BN affine, ReLU / residual, then a transpose through LDS so that every global store
    // instruction writes whole pixel rows (1 KB contiguous per wave) instead of 64 scattered 16-byte pieces.
    // Lane (m, hi) of accumulator (i, j) holds channels 32*j + 16*hi + r (r = 0..15) of pixel 32*i + m of its row.
    // Scratch: this wave's 8 KB of the halo buffer that is free until the next unit's tap-0 barrier
    // (row = one pixel = COUT*2 bytes, 16-byte slot index XORed with the pixel number: conflict-free both ways).
    const int ct = c_ct, rb = c_rb;
    const bool eg = GRP && c_b >= a.B;                               // (GRP) problem 1: image index from 0, pointers + deltas
    const int b = eg ? c_b - a.B : c_b;
    const int e_ph = c_ph;                                           // (PH) phase of the tile just computed
    if constexpr (PH) {
      // next list entry: the following phase of this tile, then the next tile; its tap-set side becomes the current one, and the
      // side of the entry after it is looked up for the pre-read at the end of the next tile
      if (++c_ph == nph) { c_ph = 0; tile_advance(c_ct, c_rb, c_b); }
      const int sn_ = side2(c_ph), sq_ = side2(c_ph + 1 == nph ? 0 : c_ph + 1);
      pcol[0] = sn_ ? aoff[1] : aoff[0]; pcol[1] = sn_ ? aoff[2] : aoff[1];
      pq0 = sq_ ? aoff[1] : aoff[0];
    } else
    tile_advance(c_ct, c_rb, c_b);                                   // (for the next iteration)
    const int oh0 = rb * C3_TH + RW * wave;                          // fragment i: output row oh0 + i / FC, columns 32*(i % FC) ..
    // opaque copies of the lane coordinates: without them every per-lane epilogue address is hoisted out of the tile loop
    // and kept (spilled) across the whole MFMA phase
    int em = m, ehi = hi, el = lane;
    if constexpr (M16) {   // (m and hi are not otherwise live in this form: derive them from the lane id instead of carrying two registers)
      asm volatile("" : "+v"(el));
      em = el & 31; ehi = el >> 5;
    } else
    asm volatile("" : "+v"(em), "+v"(ehi), "+v"(el));
    // The transpose scratch of a wave is a quarter of the free halo buffer: 10 KB (FPW 4) or 6 KB (FPW 2).  A fragment's 32
    // pixels x CW channels x 2 B must fit: all COUT channels in one pass, or (cout 128 on the 4-row tile) two passes of 64.
    constexpr int JW = (32 * COUT * 2 <= C3_HALO / 4) ? NCT : NCT / 2, CW = JW * 32, NPASS = NCT / JW;
    static_assert(32 * CW * 2 <= C3_HALO / 4 && (!HEAD || NPASS == 1 || FPW == 2), "transpose scratch");
    constexpr int ROWB = CW * 2, SPR = CW / 8, RPI = 64 / SPR;   // row bytes, 16-B slots per row, rows per store instr
    unsigned char* scr = smem + hprev(abuf) + wave * (C3_HALO / 4);   // (the buffer of the unit just consumed)
    bf16_t* __restrict__ yrow0 = a.y + (eg ? a.g_y : 0) + (size_t)b * a.y_bs + (size_t)oh0 * a.Wo * a.y_cs + a.y_co + (PH ? e_ph * a.y_pc : 0);
    // (base of image b, not of the wave's first row: rows past the image bottom must not even form an address beyond the buffer)
    const bf16_t* __restrict__ rimg0 = a.res + (eg ? a.g_res : 0) + (size_t)b * a.r_bs + a.r_co + (PH ? e_ph * a.r_pc : 0);
    // fused output conv of the tile's problem
    const unsigned char* __restrict__ e_hw = a.hw + (eg ? a.g_hw : 0);
    const float* __restrict__ e_hb = a.hb + (eg ? a.g_hb : 0);
    float* __restrict__ e_ho = a.ho + (eg ? a.g_ho : 0);
    const long e_ho_bs = a.ho_bs + (eg ? a.g_ho_bs : 0);
    const int e_hn = a.hn + (eg ? a.g_hn : 0);
    const int sh = a.sw - 1;   // stride 2: shift by 1, keep even columns
    if constexpr (SC) {
      // projection shortcut: B operand = this wave's pixels of the block input straight from global memory (lane (m, hi)
      // of k-step ks: channels 16*ks + 8*hi .. +8 of its pixel, one 16-byte load), A operand = the packed weight fragment
      // (1 KB coalesced, L2 resident), accumulated onto the conv's own accumulators.  Dead pixels read pixel (0, 0).
      // (Measured in the ISA: each weight fragment request is followed by a full wait before its two MFMAs.  Batching the
      // requests per k-step or making the k-step count a compile-time constant both made hipcc spill 56 - 165 registers
      // across the MFMA phase, so the simple form stays.)
      // All four pixel fragments of the wave, four k-steps at a time (4 x 4 x 4 = 64 registers of pixels in flight): a weight
      // fragment then feeds four MFMAs instead of two, i.e. half as many exposed L2 round trips per tile.
      const bf16_t* __restrict__ sb = a.sx + (size_t)b * a.s_bs + a.s_co + 8 * ehi;
      const unsigned char* __restrict__ wq = a.scw + el * 16;
      auto shortcut = [&](auto MKc) {
        constexpr int MK = decltype(MKc)::value;            // k-steps per pass
#pragma unroll
        for (int kh = 0; kh < 8 / MK; ++kh) {
          if (kh * MK < a.s_nks) {
            s16x8 sxq[FPW][MK];
#pragma unroll
            for (int i = 0; i < FPW; ++i) {
              const int tc = 32 * (i % FC) + em, ow = ct * C3_TW + tc, oh = oh0 + i / FC;
              const bool live = tc < C3_TW && ow < a.W && oh < a.H;
              const bf16_t* sp = sb + (live ? ((size_t)oh * a.W + ow) * a.s_cs : 0) + 16 * MK * kh;
#pragma unroll
              for (int ks = 0; ks < MK; ++ks)
                if (kh * MK + ks < a.s_nks) sxq[i][ks] = *(const s16x8*)(sp + 16 * ks);
            }
#pragma unroll
            for (int ks = 0; ks < MK; ++ks)
              if (kh * MK + ks < a.s_nks) {
#pragma unroll
                for (int j = 0; j < NCT; ++j) {
                  const s16x8 wf = *(const s16x8*)(wq + (size_t)((kh * MK + ks) * NCT + j) * 1024);
#pragma unroll
                  for (int i = 0; i < FPW; ++i)
                    acc[i][j] = H16<DT>::mfma(wf, sxq[i][ks], acc[i][j]);
                }
              }
          }
          C3_FENCE();
        }
      };
      shortcut(std::integral_constant<int, 4>{});
      C3_FENCE();
    }
    // FL >= 0: the flag combination is a compile-time constant (no per-value selects); FL < 0: read a.flags
    auto epilogue = [&](auto FL) {
      constexpr int F = decltype(FL)::value;
      const bool relu_pre = F >= 0 ? (F & RD_RELU_PRE) != 0 : (a.flags & RD_RELU_PRE) != 0;
      const bool do_add = F >= 0 ? (F & RD_ADD) != 0 : (a.flags & RD_ADD) != 0;
      const bool relu_post = F >= 0 ? (F & RD_RELU_POST) != 0 : (a.flags & RD_RELU_POST) != 0;
      // residuals: the loads of pixel fragment i+1 are issued before fragment i is processed (dead pixels read pixel 0)
      Slot16 rv[2][NCT][2];
      auto res_load = [&](int i, Slot16 (&dst)[NCT][2]) {
        const int tc = 32 * (i % FC) + em, ow = ct * C3_TW + tc, oh = oh0 + i / FC;
        const bool live = tc < C3_TW && ow < a.W && oh < a.H && !(ow & sh) && !(DBG & 64);   // (DBG 64: every lane reads pixel 0 -- L2 hits)
        const bf16_t* rp = rimg0 + (live ? ((size_t)oh * a.Wo + (size_t)(ow >> sh)) * a.r_cs : 0) + 16 * ehi;
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
          if (DBG & 128) { dst[j][0] = Slot16{0, 0, 0, 0}; dst[j][1] = Slot16{0, 0, 0, 0}; continue; }   // (DBG 128: no residual load)
          dst[j][0] = *(const Slot16*)(rp + j * 32);
          dst[j][1] = *(const Slot16*)(rp + j * 32 + 8);
        }
      };
      if (do_add) res_load(0, rv[0]);
      // ReLU placement: a ReLU that is the last operation before the bf16 conversion is applied AFTER it, as a packed signed
      // 16-bit max (negative bf16 values are negative integers); only relu-then-add needs the fp32 form.
      const bool relu_f32 = relu_pre && do_add;
      const bool relu_i16 = relu_post || (relu_pre && !do_add);
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      // scale / shift of a channel block.  cout 64: both blocks are read from LDS once per tile and stay in registers;
      // cout 128 (no room for all four): read one (i, j) block ahead of its use
      constexpr bool SC_HOIST = NCT == 2;
      f32x4 scq[2][4], shq[2][4];
      auto sc_load = [&](int j, f32x4 (&sc)[4], f32x4 (&sh)[4]) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          sc[g4] = *(const f32x4*)(Sc + j * 32 + 16 * ehi + 4 * g4);
          sh[g4] = *(const f32x4*)(Sc + COUT + j * 32 + 16 * ehi + 4 * g4);
        }
      };
      if constexpr (!FOLD) {
        sc_load(0, scq[0], shq[0]);
        if constexpr (SC_HOIST) sc_load(1, scq[1], shq[1]);
      }
#pragma unroll
      for (int i = 0; i < FPW; ++i) {
        if (do_add && i + 1 < FPW) res_load(i + 1, rv[(i + 1) & 1]);
        f32x16 h0, h1;                       // HEAD: the output conv's accumulators (weights hi / lo), summed over the passes
        if constexpr (HEAD) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { h0[r] = 0.f; h1[r] = 0.f; }
        }
#pragma unroll
        for (int jp = 0; jp < NPASS; ++jp) {
        // HEAD on two passes: this pass's eight weight fragments (k-steps 4*jp .., hi and lo) straight from global memory
        // (16 KB, L2 resident), requested before the pass's conversion work so that their latency is hidden behind it
        constexpr int KSP = 8 / NPASS;
        s16x8 hwq[HEAD && NPASS == 2 ? 2 : 1][HEAD && NPASS == 2 ? KSP : 1];
        if constexpr (HEAD && NPASS == 2) {
#pragma unroll
          for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int ks = 0; ks < KSP; ++ks)
              hwq[part][ks] = *(const s16x8*)(e_hw + part * 8192 + (jp * KSP + ks) * 1024 + el * 16);
        }
#pragma unroll
        for (int j = jp * JW; j < (jp + 1) * JW; ++j) {
          C3_FENCE();   // one (i, j) accumulator at a time: keeps the register footprint of the epilogue small
          const int cur = SC_HOIST ? j : (i * NCT + j) & 1;
          if constexpr (!SC_HOIST && !FOLD) sc_load((j + 1) % NCT, scq[cur ^ 1], shq[cur ^ 1]);
          const int cb = j * 32 + 16 * ehi;
          unsigned pk[8];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
              const int r = 4 * g4 + 2 * h2;
              const f32x2 av = {acc[i][j][r], acc[i][j][r + 1]};
              f32x2 v = av;                                            // FOLD: scale in the weights, shift in the accumulator
              if constexpr (!FOLD) {
                const f32x2 s2 = {scq[cur][g4][2 * h2], scq[cur][g4][2 * h2 + 1]};
                const f32x2 t2 = {shq[cur][g4][2 * h2], shq[cur][g4][2 * h2 + 1]};
                v = av * s2 + t2;                                      // v_pk_fma_f32
              }
              if (relu_f32) v = f32x2{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)};
              if (do_add) {
                const unsigned w2 = rv[i & 1][j][r >> 3][(r >> 1) & 3];
                v += H16<DT>::unpk(w2);
              }
              unsigned p2 = H16<DT>::pk(v[0], v[1]);
              if (relu_i16) p2 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p2), (s16x2){0, 0}));
              pk[2 * g4 + h2] = p2;
            }
          }
#pragma unroll
          for (int u = 0; u < 2; ++u)
            *(Slot16*)(scr + em * ROWB + (((((cb - jp * CW) >> 3) + u) ^ (em & (SPR - 1))) << 4)) =
                Slot16{pk[4 * u], pk[4 * u + 1], pk[4 * u + 2], pk[4 * u + 3]};
        }
        C3_FENCE();
        __builtin_amdgcn_wave_barrier();   // (the wave runs in lockstep on hardware; this orders the lanes under hipemu)
        if constexpr (HEAD) {
          // out[o][px] = sum_c w[o][c] * act[px][c]: A = weights (hi, lo), B = the fragment's pixels from the scratch image
#pragma unroll
          for (int ks = 0; ks < KSP; ++ks) {
            const s16x8 bq = *(const s16x8*)(scr + em * ROWB + (((2 * ks + ehi) ^ (em & (SPR - 1))) << 4));
            s16x8 wh, wl;
            if constexpr (NPASS == 2) { wh = hwq[0][ks]; wl = hwq[1][ks]; }
            else {
              wh = *(const s16x8*)(smem + HWOFF + ks * 1024 + el * 16);
              wl = *(const s16x8*)(smem + HWOFF + 8192 + ks * 1024 + el * 16);
            }
            h0 = H16<DT>::mfma(wl, bq, h0);
            h1 = H16<DT>::mfma(wh, bq, h1);
          }
          const int tcs = 32 * (i % FC) + em, ows = ct * C3_TW + tcs, ohs = oh0 + i / FC;
          if (jp == NPASS - 1 && tcs < C3_TW && ows < a.W && ohs < a.H) {
            float* o = e_ho + (size_t)b * e_ho_bs + (a.ho_off + (size_t)ohs * a.W + ows) * e_hn + 4 * ehi;
            if (e_hn == 8 && !((size_t)o & 15)) {   // (the box regression head: one 16-byte store per lane instead of four 4-byte ones)
              const f32x4 bv = *(const f32x4*)(e_hb + 4 * ehi);
              *(f32x4*)o = f32x4{(h0[0] + h1[0]) + bv[0], (h0[1] + h1[1]) + bv[1], (h0[2] + h1[2]) + bv[2], (h0[3] + h1[3]) + bv[3]};
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (4 * ehi + r < e_hn) o[r] = (h0[r] + h1[r]) + e_hb[4 * ehi + r];
            }
          }
        } else {
          // read back pixel-major and store: lane -> (pixel it*RPI + el / SPR, slot el % SPR)
#pragma unroll
          for (int it = 0; it < 32 / RPI; ++it) {
            const int pr = it * RPI + el / SPR, sl = el % SPR;
            const Slot16 v = *(const Slot16*)(scr + pr * ROWB + ((sl ^ (pr & (SPR - 1))) << 4));
            const int tcs = 32 * (i % FC) + pr, ows = ct * C3_TW + tcs;
            if (tcs < C3_TW && ows < a.W && oh0 + i / FC < a.H && !(ows & sh) && (!(DBG & 1) || a.B < 0))
              // non-temporal: an output line is written once and read back by the NEXT launch; without the hint the stores displace the
              // halo rows / residual lines that neighbouring tiles are about to re-read (A/B on one box, 7 alternations: +0.7 ... +0.9 %
              // frames/s, serial step sum -2.5 ... -3.3 %; a run-time size test around the store gave the gain away again, EXPERIMENTS.md)
              __builtin_nontemporal_store(v, (Slot16*)(yrow0 + (size_t)(i / FC) * a.Wo * a.y_cs + (size_t)(ows >> sh) * a.y_cs + jp * CW + sl * 8));
          }
        }
        __builtin_amdgcn_wave_barrier();
        C3_FENCE();
        }   // channel pass
      }
    };
    // M16: lane (n, q) of accumulator (h = 2*row + half, cb) holds channels 32*(cb/2) + 8*q + 4*(cb & 1) + r (r = 0..3) of pixel
    // 16*half + n of its row, i.e. per fragment pair the 8 contiguous channels of ONE 16-byte slot: convert, (add the residual slot,)
    // clamp, write the slot into the same pixel-major scratch image, read it back and store as above.
    auto epilogue16 = [&](auto FL) {
      constexpr int F = decltype(FL)::value;
      const bool relu_pre = F >= 0 ? (F & RD_RELU_PRE) != 0 : (a.flags & RD_RELU_PRE) != 0;
      const bool do_add = F >= 0 ? (F & RD_ADD) != 0 : (a.flags & RD_ADD) != 0;
      const bool relu_post = F >= 0 ? (F & RD_RELU_POST) != 0 : (a.flags & RD_RELU_POST) != 0;
      const bool relu_f32 = relu_pre && do_add;
      const bool relu_i16 = relu_post || (relu_pre && !do_add);
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      const int en = el & 15, eq = el >> 4;
      Slot16 rv[2][2][JW];                                   // [buffer][pixel half][fragment pair of the pass]: one pass (row i, 64 channels) ahead
      auto res_load = [&](int t, Slot16 (&dst)[2][JW]) {      // pass t = NPASS * i + jp
        const int i = t / NPASS, jp = t % NPASS;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int ow = ct * C3_TW + 16 * hf + en, oh = oh0 + i;
          const bool live = ow < a.W && oh < a.H;
          const bf16_t* rp = rimg0 + (live ? ((size_t)oh * a.Wo + (size_t)ow) * a.r_cs : 0) + 8 * eq + jp * CW;
#pragma unroll
          for (int pp = 0; pp < JW; ++pp) dst[hf][pp] = *(const Slot16*)(rp + 32 * pp);
        }
      };
      if (do_add) res_load(0, rv[0]);
#pragma unroll
      for (int i = 0; i < FPW; ++i) {
        // HEAD: the 1x1 output conv as 16 x 16 x 32 MFMAs (nout <= 8 rows of 16) whose B operand is the slot this lane has just rounded --
        // lane (n, q) holds the 8 channels 32*p + 8*q .. of pixel 16*half + n, which is lane (n, q)'s share of the k-step of pair p
        // (pack_head_frag16) -- so the 128-channel result goes neither to HBM nor through LDS.  Accumulators [pixel half][weights lo / hi].
        f32x4 hh[HEAD ? 2 : 1][HEAD ? 2 : 1];
        if constexpr (HEAD) {
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int part = 0; part < 2; ++part) hh[hf][part] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int jp = 0; jp < NPASS; ++jp) {
          const int t = NPASS * i + jp;
          if (do_add && t + 1 < NPASS * FPW) res_load(t + 1, rv[(t + 1) & 1]);
          // HEAD: this pass's weight fragments (hi and lo of pairs 2*jp, 2*jp + 1: L2 resident), requested before the conversion work
          s16x8 hwq[HEAD ? 2 : 1][HEAD ? JW : 1];
          if constexpr (HEAD) {
#pragma unroll
            for (int part = 0; part < 2; ++part)
#pragma unroll
              for (int pp = 0; pp < JW; ++pp)
                hwq[part][pp] = *(const s16x8*)(e_hw + part * 4096 + (jp * JW + pp) * 1024 + el * 16);
          }
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int pp = 0; pp < JW; ++pp) {
              C3_FENCE();   // one 16-byte slot at a time: keeps the register footprint of the epilogue small
              const int p = jp * JW + pp, h = 2 * i + hf;
              unsigned pk[4];
#pragma unroll
              for (int w2 = 0; w2 < 4; ++w2) {
                const f32x4 av = acc6[h][2 * p + (w2 >> 1)];
                f32x2 v = {av[2 * (w2 & 1)], av[2 * (w2 & 1) + 1]};
                if (relu_f32) v = f32x2{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)};
                if (do_add) v += H16<DT>::unpk(rv[t & 1][hf][pp][w2]);
                unsigned p2 = H16<DT>::pk(v[0], v[1]);
                if (relu_i16) p2 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p2), (s16x2){0, 0}));
                pk[w2] = p2;
              }
              if constexpr (HEAD) {
                s16x8 bq;
                memcpy(&bq, pk, 16);
                hh[hf][0] = H16<DT>::mfma16(hwq[1][pp], bq, hh[hf][0]);
                hh[hf][1] = H16<DT>::mfma16(hwq[0][pp], bq, hh[hf][1]);
              } else {
                // (the lane's 16-byte slot straight to global memory instead -- 64-byte runs per pixel, no LDS round trip -- measured slower:
                //  W 2656 312 -> 331 us, profiles/r06l_conv16_direct_store_ab.txt)
                const int px = 16 * hf + en;
                *(Slot16*)(scr + px * ROWB + (((4 * pp + eq) ^ (px & (SPR - 1))) << 4)) = Slot16{pk[0], pk[1], pk[2], pk[3]};
              }
            }
          C3_FENCE();
          if constexpr (!HEAD) {
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int it = 0; it < 32 / RPI; ++it) {
            const int pr = it * RPI + el / SPR, sl = el % SPR;
            const Slot16 v = *(const Slot16*)(scr + pr * ROWB + ((sl ^ (pr & (SPR - 1))) << 4));
            const int ows = ct * C3_TW + pr;
            if (ows < a.W && oh0 + i < a.H)
              __builtin_nontemporal_store(v, (Slot16*)(yrow0 + (size_t)i * a.Wo * a.y_cs + (size_t)ows * a.y_cs + jp * CW + sl * 8));
          }
          __builtin_amdgcn_wave_barrier();
          C3_FENCE();
          }
        }
        if constexpr (HEAD) {   // out[o][px]: lane (n, q) holds outputs 4*q .. 4*q + 3 (q < 2: nout <= 8) of pixel 16*half + n
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const int ows = ct * C3_TW + 16 * hf + en, ohs = oh0 + i;
            if (eq < 2 && ows < a.W && ohs < a.H) {
              float* o = e_ho + (size_t)b * e_ho_bs + (a.ho_off + (size_t)ohs * a.W + ows) * e_hn + 4 * eq;
              if (e_hn == 8 && !((size_t)o & 15)) {   // (the box regression head: one 16-byte store per lane)
                const f32x4 bv = *(const f32x4*)(e_hb + 4 * eq);
                *(f32x4*)o = f32x4{(hh[hf][0][0] + hh[hf][1][0]) + bv[0], (hh[hf][0][1] + hh[hf][1][1]) + bv[1],
                                   (hh[hf][0][2] + hh[hf][1][2]) + bv[2], (hh[hf][0][3] + hh[hf][1][3]) + bv[3]};
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (4 * eq + r < e_hn) o[r] = (hh[hf][0][r] + hh[hf][1][r]) + e_hb[4 * eq + r];
              }
            }
          }
          C3_FENCE();
        }
      }
    };
    if constexpr (M16) {
      static_assert(!M16 || (NPASS == 2 && JW == 2 && CW == 64), "M16 epilogue: two passes of 64 channels");
      if (a.flags == RD_RELU_POST) epilogue16(std::integral_constant<int, RD_RELU_POST>{});
      else if (a.flags == (RD_ADD | RD_RELU_POST)) epilogue16(std::integral_constant<int, RD_ADD | RD_RELU_POST>{});
      else epilogue16(std::integral_constant<int, -1>{});
    } else {
    if (a.flags == RD_RELU_POST) epilogue(std::integral_constant<int, RD_RELU_POST>{});
    else if (a.flags == (RD_ADD | RD_RELU_POST)) epilogue(std::integral_constant<int, RD_ADD | RD_RELU_POST>{});
    else if (TS != 0 && a.flags == (RD_RELU_PRE | RD_ADD)) epilogue(std::integral_constant<int, RD_RELU_PRE | RD_ADD>{});   // skip + relu(BN(deconv))
    else epilogue(std::integral_constant<int, -1>{});
    }
    if (k == 0) C3_TRACE()
  }
  // DMA still in flight (the dummy tail fetches) targets this workgroup's LDS: retire it before the LDS is released
  __builtin_amdgcn_s_waitcnt(RD_VMCNT_IMM(0));
  C3_TRACE()
  if (a.trace && tid == 0) a.trace[(size_t)blockIdx.x * 8 + 7] = __builtin_readcyclecounter() - clk0;   // shader-clock ticks of the whole life
#undef C3_STEP16
#undef C3_MM16
#undef C3_RDP16
#undef C3_RDW16
#undef C3_GBODY_FROM1
#undef C3_RDX
#undef C3_GSTEP
#undef C3_AO
#undef C3_STEP
#undef C3_STEP_
#undef C3_MMZ
#undef C3_SYNC
#undef C3_MM
#undef C3_RD
#undef C3_FENCE
#undef C3_TRACE
}

// Build options (none is set by rangedet_amd.build; tests/emu/build_emu.sh passes both to keep the CPU emulation of the persistent
// kernels small and its compile time in minutes):
//   -DRD_BUILD_NUM_CUS=n                     workgroup slots are sized for n compute units instead of asking the device
//   -DRD_BUILD_F16_PRODUCTION_FORMS_ONLY     fp16: only the launch forms the lowering emits (folded scales on the two-workgroup
//                                            tiles, fused output conv); other fp16 shapes take the generic tap kernel
inline int conv_num_cus() {
#ifdef RD_BUILD_NUM_CUS
  return RD_BUILD_NUM_CUS;
#else
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  return v;
#endif
}

// One launch of one instantiation; the first launch of each raises its dynamic-LDS limit (once per process and instantiation).
template <int NCT, int TS, bool HEAD, bool SC, bool FOLD, int FPW, int FC, int NHB, int DT, bool WD = false, bool GRP = false, int BODY = 0, bool M16 = false>
inline int c3_go(int grid, hipStream_t st, const Conv3Args& a) {
  auto k = conv3x3_stream_kernel<NCT, 0, TS, HEAD, SC, FOLD, FPW, FC, NHB, DT, WD, GRP, BODY, M16>;
  static std::atomic<unsigned long long> seen{0};
  once_per_device(seen, [&] { allow_big_lds(k); });
  constexpr size_t lds = C3Cfg<NCT, FPW, FC, NHB, WD>::LDS + (HEAD && FPW == 4 ? 16384 : 0);   // (8 x 62 tile: the output conv's weights in LDS)
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, a);
  return check_launch("conv3x3_stream_kernel");
}

// (RD_BUILD_F16_PRODUCTION_FORMS_ONLY, see above; the GPU library has every form in both 16-bit types)
#ifdef RD_BUILD_F16_PRODUCTION_FORMS_ONLY
constexpr bool kF16AllForms = false;
#else
constexpr bool kF16AllForms = true;
#endif
inline bool conv3_has_form(int dt, bool fold) { return kF16AllForms || dt == RD_BF16 || fold; }

inline bool conv3_eligible(const TapList& tl, int in_stride, int out_stride, int cout, int dt, int Win, int Wq, int Wout) {
  if (!is_h16(dt) || !conv3_has_form(dt, false) || tl.n != 9 || (in_stride != 1 && in_stride != 2) || out_stride != 1) return false;
  if (Wq != (Win - 1) / in_stride + 1 || Wout != Wq) return false;   // pad 1, kernel 3
  if (cout != 64 && cout != 128) return false;
  for (int t = 0; t < 9; ++t)
    if (tl.dh[t] != t / 3 - 1 || tl.dw[t] != t % 3 - 1) return false;
  return !dev_switches().conv_v1;
}

// Second problem of a two-problem launch (GRP): same shapes, channel strides and flags as the first, its own tensors.
struct Conv3Second {
  const void* x; int x_co;
  const void* w; const float* shift;
  const void* res; int r_co;
  void* y; int y_co;
  const void* hw; const float* hb; float* ho; long ho_bs; int hn;   // fused output conv (when the first problem has one)
};
// All phases of a transposed conv in one launch (PH form, TS = 3): per-phase weight images w + ph * w_pb, tap-set side per phase
struct Conv3Phases { int nph, ts_mask, y_pc, r_pc; long w_pb; };
// Second input tensor of a conv over a channel concatenation [x | x2] (Conv3Args::x2): cin of the launch = cin1 + cin2 where cin1
// (a multiple of 32) channels come from x
struct Conv3Src2 { const void* x; int cs, co, cin1, cin2; };
inline bool conv3_phases_eligible(int cout, int flags) {
  const DevSwitches& sw_ = dev_switches();
  return (cout == 64 || cout == 128) && (flags & RD_SCALE_FOLDED) && sw_.conv_w30 && sw_.conv_wide && !sw_.conv_v1 &&
         ((cout == 64 && sw_.conv_th4 != 3) || (cout == 128 && sw_.conv_th4 && sw_.conv_w30 == 2));
}
// Does the launcher have the v_mfma_f32_16x16x32 form (RD_MFMA16) for this conv?  The packer side of the contract: an image made by
// rd_pack_conv3x3_m16_host may only be launched where this says yes (the launch fails loudly otherwise).
inline bool conv3_mfma16_ok(int cin, int cout, int stride_w, int W, bool headfuse) {
  const DevSwitches& sw_ = dev_switches();
  const bool head30_ok = sw_.conv_w30 == 2 && sw_.conv_head30 != 0;
  return cout == 128 && stride_w == 1 && cin >= 32 && cin % 32 == 0 && !sw_.conv_v1 && sw_.conv_wide && sw_.conv_w30 == 2 && sw_.conv_th4 &&
         (sw_.conv_th4 != 2 || W >= 600) && (!headfuse || head30_ok);
}
inline bool conv3_pair_eligible(int cout, int flags, int W, bool headfuse = false) {
  const DevSwitches& sw_ = dev_switches();
  return cout == 128 && (flags & RD_SCALE_FOLDED) && sw_.conv_th4 && sw_.conv_w30 == 2 && sw_.conv_wide && (sw_.conv_th4 != 2 || W >= 600) &&
         (!headfuse || sw_.conv_head30 != 0);   // (RD_CONV_HEAD30=0: no fused output conv on the two-workgroup tiles -> two launches)
}

template <int DT>
inline int launch_conv3_dt(const void* x, int x_cs, int x_co, const void* w, const float* scale, const float* shift,
                           const void* res, int r_cs, int r_co, void* y, int y_cs, int y_co, int B, int H, int W, int cin,
                           int cout, int flags, int sw, hipStream_t st, int ts, const Conv3Args* head, const Conv3Second* g1,
                           const Conv3Phases* ph, const Conv3Src2* s2, int body);
inline int launch_conv3(const void* x, int x_cs, int x_co, const void* w, const float* scale, const float* shift,
                        const void* res, int r_cs, int r_co, void* y, int y_cs, int y_co, int B, int H, int W, int cin,
                        int cout, int flags, int sw, hipStream_t st, int ts, const Conv3Args* head, int dt,
                        const Conv3Second* g1, const Conv3Phases* ph, const Conv3Src2* s2, int body) {
  RD_REQUIRE(is_h16(dt), RD_EINVAL, "conv3: dtype %d (the persistent 3x3 kernel takes RD_BF16 or RD_F16)", dt);
  if (dt == RD_F16) return launch_conv3_dt<RD_F16>(x, x_cs, x_co, w, scale, shift, res, r_cs, r_co, y, y_cs, y_co, B, H, W, cin, cout, flags, sw, st, ts, head, g1, ph, s2, body);
  return launch_conv3_dt<RD_BF16>(x, x_cs, x_co, w, scale, shift, res, r_cs, r_co, y, y_cs, y_co, B, H, W, cin, cout, flags, sw, st, ts, head, g1, ph, s2, body);
}

template <int DT>
inline int launch_conv3_dt(const void* x, int x_cs, int x_co, const void* w, const float* scale, const float* shift,
                           const void* res, int r_cs, int r_co, void* y, int y_cs, int y_co, int B, int H, int W, int cin,
                           int cout, int flags, int sw, hipStream_t st, int ts, const Conv3Args* head, const Conv3Second* g1,
                           const Conv3Phases* ph, const Conv3Src2* s2, int body) {
  Conv3Args a;
  memset(&a, 0, sizeof(a));
  a.ngrp = g1 ? 2 : 1;
  a.nph = 1;
  if (ph) { a.nph = ph->nph; a.ts_mask = ph->ts_mask; a.y_pc = ph->y_pc; a.r_pc = ph->r_pc; a.w_pb = ph->w_pb; }
  if (head) { a.hw = head->hw; a.hb = head->hb; a.ho = head->ho; a.ho_bs = head->ho_bs; a.ho_off = head->ho_off; a.hn = head->hn; }
  const bool sc = head && head->sx;
  const bool fold = (flags & RD_SCALE_FOLDED) != 0;
  RD_REQUIRE(!fold || !scale, RD_EINVAL, "conv3: RD_SCALE_FOLDED takes no scale array (the packer folded it into the weights)");
  flags &= ~RD_SCALE_FOLDED;
  if (sc) { a.sx = head->sx; a.s_cs = head->s_cs; a.s_co = head->s_co; a.s_bs = head->s_bs; a.scw = head->scw; a.s_nks = head->s_nks; }
  a.sw = sw; a.Wo = (W - 1) / sw + 1;
  a.x = (const bf16_t*)x; a.x_cs = x_cs; a.x_co = x_co; a.x_bs = (long)H * W * x_cs;
  a.w = (const unsigned char*)w; a.scale = scale; a.shift = shift;
  a.res = (const bf16_t*)res; a.r_cs = r_cs; a.r_co = r_co; a.r_bs = (long)H * a.Wo * r_cs;
  a.y = (bf16_t*)y; a.y_cs = y_cs; a.y_co = y_co; a.y_bs = (long)H * a.Wo * y_cs;
  // zero bytes for padding: the tail every packer appends to the weight image (k_conv.h RD_CONV_TAIL)
  a.zero16 = (const unsigned char*)w + conv_packed_body_bytes(c3_nsteps(ts), cin, cout, RD_BF16);
  a.H = H; a.W = W; a.B = B; a.nslots = cin_slots(cin, RD_BF16); a.nchunk = (cin + 31) / 32; a.flags = flags;   // (without RD_SCALE_FOLDED)
  // Two workgroups per CU (FPW 2) for every layer with folded scale and no fused output conv, on 8-row x 30-column tiles
  // (FC 1).  Measured A/B, frames/s end to end: 4 x 62 tiles for the cout-128 layers +2.5 % over 8 x 62 everywhere (W = 332
  // layers 70 -> 53 us, W = 1328 head convs 200 -> 179 us); 8 x 30 for the cout-64 layers +0.7 % on top (the HBM-bound
  // layers keep their 10/8 halo rows); 8 x 30 instead of 4 x 62 for the cout-128 layers another +1.4 % (14 % fewer halo
  // pieces per pixel, ring depth 4 instead of 3).  Dev switches: RD_CONV_TH4=0 -> cout 128 on 8 x 62, =3 -> cout 64 on
  // 4 x 62 too; RD_CONV_W30=0 -> no 8 x 30 tiles, =1 -> only for cout 64.
  const DevSwitches& sw_ = dev_switches();
  const int th4_mode = sw_.conv_th4, head30 = sw_.conv_head30, w30_mode = sw_.conv_w30;
  const bool hb3 = sw_.conv_hb3 != 0;   // cout 64 on the 8 x 30 tiles: halo fetch two units ahead (three buffers)
  const bool headfuse = head && !sc;
  // (fused output conv on the two-workgroup tiles: its 16 KB of weights no longer fit in LDS and are re-read from L2 per
  //  fragment and pass; measured per layer: W = 1328 213 -> 205 us, W = 664 109 -> 107 us, W = 2656 398 -> 405 us, so the
  //  full-width level stays on the 8 x 62 tile.  RD_CONV_HEAD30=0 -> never, =2 -> always)
  // (a two-problem launch always takes the two-workgroup tile: its output-conv weights are per problem and come from L2)
  // (round 6: in the 16 x 16 x 32 form the two-workgroup tile wins at full width too -- W 2656: 352 -> 328 us, +0.5 % frames/s on one box,
  //  profiles/EXPERIMENTS.md -- so an RD_MFMA16 image always takes it)
  const bool head30_ok = w30_mode == 2 && (head30 == 2 || (head30 == 1 && (W <= 1400 || body == C3_BODY_M16)) || (g1 && head30));   // (only exists on the 8 x 30 tiles)
  const bool th4 = th4_mode && (cout == 128 || th4_mode == 3) && fold && (!headfuse || head30_ok) && (th4_mode != 2 || W >= 600);
  const bool w30_128 = w30_mode == 2 && th4 && cout == 128;
  const bool w30 = w30_mode && fold && ((!th4 && cout == 64 && !headfuse) || w30_128);
  RD_REQUIRE(!headfuse || !th4 || w30, RD_EINVAL, "conv3 + output conv: two workgroups per CU only on the 8 x 30 tiles");
  const bool wd = sw_.conv_wide && w30;   // 8 x 32 tiles (34-pixel halo pitch) instead of 8 x 30: no discarded MFMA columns
  const int th = th4 && !w30 ? 4 : C3_TH, tw = wd ? 32 : w30 ? C3Cfg<2, 2, 1>::TW : C3_TW;
  a.ncol = (W + tw - 1) / tw; a.nrow = (H + th - 1) / th; a.ntiles = a.ncol * a.nrow * B * a.ngrp;
  if (g1) {
    RD_REQUIRE(wd && fold && !sc && ts == 0 && cout == 128 && sw == 1, RD_ESHAPE,
               "conv3: two problems per launch need the 8 x 32 tile form (cout 128, folded scales, stride 1, no shortcut)");
    RD_REQUIRE(!headfuse == !g1->hw && !res == !g1->res && !shift == !g1->shift, RD_EINVAL,
               "conv3: the two problems of a launch must have the same structure (output conv / residual / shift)");
    a.g_x = ((const bf16_t*)g1->x + g1->x_co) - ((const bf16_t*)x + x_co);
    a.g_w = (const unsigned char*)g1->w - (const unsigned char*)w;
    a.g_shift = shift ? g1->shift - shift : 0;
    a.g_res = res ? ((const bf16_t*)g1->res + g1->r_co) - ((const bf16_t*)res + r_co) : 0;
    a.g_y = y ? ((bf16_t*)g1->y + g1->y_co) - ((bf16_t*)y + y_co) : 0;
    if (headfuse) {
      a.g_hw = (const unsigned char*)g1->hw - a.hw; a.g_hb = g1->hb - a.hb; a.g_ho = g1->ho - a.ho;
      a.g_ho_bs = g1->ho_bs - a.ho_bs; a.g_hn = g1->hn - a.hn;
    }
  }
  if (s2) {   // conv over [x | x2]: cin = cin1 + cin2, the first cin1 (full 32-channel chunks) from x
    RD_REQUIRE(wd && !g1 && sw == 1 && s2->cin1 > 0 && s2->cin1 % 32 == 0 && s2->cin1 + s2->cin2 == cin && s2->cin2 > 0, RD_ESHAPE,
               "conv3: a two-tensor input needs the 8 x 32 tile form and cin1 a multiple of 32 (cin1 %d, cin2 %d, cin %d)", s2->cin1, s2->cin2, cin);
    a.x2 = (const bf16_t*)s2->x; a.x2_cs = s2->cs; a.x2_co = s2->co; a.x2_bs = (long)H * W * s2->cs;
    a.nchunk1 = s2->cin1 / 32; a.nslots2 = cin_slots(s2->cin2, RD_BF16); a.nslots = a.nchunk1 * 4;
  }
  const int grid = std::min(a.ntiles, conv_num_cus() * (th4 || w30 ? 2 : 1));
  a.xcd = sw_.conv_xcd && !g1 && (a.ncol * B) % 8 == 0 && grid % 8 == 0;   // (a pure permutation of the tile list under these conditions)
  if (conv_trace_buf() && (size_t)grid * 8 <= (1u << 20)) a.trace = conv_trace_buf();
  ProfScope ps(RD_PROF_CONV3, st);
#ifdef RD_CONV3_DEV   // ablation variants (DBG bits: 2 no barrier, 4 no DMA after the prologue, 16 halo from the zero page, 32 no vmcnt wait)
  static const int dbg = getenv("RD_CONV3_DBG") ? atoi(getenv("RD_CONV3_DBG")) : 0;
#define C3_DBG_CASE(D) if (cout == 128 && dbg == D) { (void)hipFuncSetAttribute((const void*)conv3x3_stream_kernel<4, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL((conv3x3_stream_kernel<4, D>), dim3(grid), dim3(256), C3Cfg<4>::LDS, st, a); return check_launch("conv3x3_stream_kernel"); }
  C3_DBG_CASE(2) C3_DBG_CASE(4) C3_DBG_CASE(16) C3_DBG_CASE(32)
#undef C3_DBG_CASE
  // cout 64 on the 8 x 30 tiles (plain 3x3, folded scales): 16 halo from the zero page, 64 residual from one L2-resident pixel,
  // 128 no residual load, 4 no DMA after the prologue, 8 no MFMAs, 256 halo from the first MB of x (L2-resident REAL data), 1 no stores
#define C3_DBG64(D)                                                                                                     \
  if (cout == 64 && w30 && !sc && ts == 0 && dbg == D) {                                                                \
    if (hb3) { auto k = conv3x3_stream_kernel<2, D, 0, false, false, true, 2, 1, 3>; allow_big_lds(k);                  \
               hipLaunchKernelGGL(k, dim3(grid), dim3(256), (C3Cfg<2, 2, 1, 3>::LDS), st, a); }                         \
    else { auto k = conv3x3_stream_kernel<2, D, 0, false, false, true, 2, 1, 2>; allow_big_lds(k);                      \
           hipLaunchKernelGGL(k, dim3(grid), dim3(256), (C3Cfg<2, 2, 1, 2>::LDS), st, a); }                             \
    return check_launch("conv3x3_stream_kernel<dbg>");                                                                  \
  }
  C3_DBG64(4) C3_DBG64(8) C3_DBG64(16) C3_DBG64(64) C3_DBG64(128) C3_DBG64(256) C3_DBG64(257)
#undef C3_DBG64
#define C3_DBG128(D)                                                                                                    \
  if (cout == 128 && w30 && !sc && !headfuse && ts == 0 && dbg == D) {                                                  \
    auto k = conv3x3_stream_kernel<4, D, 0, false, false, true, 2, 1, 2>; allow_big_lds(k);                             \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), (C3Cfg<4, 2, 1, 2>::LDS), st, a);                                      \
    return check_launch("conv3x3_stream_kernel<dbg>");                                                                  \
  }
  C3_DBG128(4) C3_DBG128(8) C3_DBG128(16) C3_DBG128(2) C3_DBG128(32) C3_DBG128(256) C3_DBG128(257)
#undef C3_DBG128
#endif
#ifdef RD_CONV3_DEV_M16_ONLY   // tools/micro/conv16_dev.hip: only the plain cout-128 form on the 8 x 32 tiles, in its two MFMA shapes (a one-minute build)
  RD_REQUIRE(wd && fold && !sc && !g1 && !ph && !s2 && sw == 1 && ts == 0 && cout == 128 && DT == RD_BF16 && (!headfuse || body == C3_BODY_M16), RD_ESHAPE, "conv16_dev: plain bf16 cout-128 form only");
  if (body == C3_BODY_M16 && headfuse) return c3_go<4, 0, true, false, true, 2, 1, 2, RD_BF16, true, false, 0, true>(grid, st, a);
  if (body == C3_BODY_M16) return c3_go<4, 0, false, false, true, 2, 1, 2, RD_BF16, true, false, 0, true>(grid, st, a);
  return c3_go<4, 0, false, false, true, 2, 1, 2, RD_BF16, true>(grid, st, a);
#else
  constexpr bool kAllForms = kF16AllForms || DT == RD_BF16;
  RD_REQUIRE(kAllForms || (fold && w30 && (cout == 128 || hb3 || sw_.conv_wide)) || (headfuse && fold), RD_ESHAPE,
             "conv3: this fp16 launch form is not instantiated in the emulator build (conv3_has_form)");
  if (sc) {
    RD_REQUIRE(ts == 0 || ts == 1, RD_ESHAPE, "conv3 + shortcut: tap set %d", ts);
    RD_REQUIRE(fold, RD_EINVAL, "conv3 + shortcut: the weights must carry the folded scales (RD_SCALE_FOLDED)");
  }
  // (tile shape, cout) -> instantiation; within it (tap set, shortcut).  All of these carry folded scales.
#define C3_BODY(N, FPW_, FC_, NHB_) C3_BODY_(N, FPW_, FC_, NHB_, false)
#define C3_BODY_(N, FPW_, FC_, NHB_, WD_)                                                                 \
  {                                                                                                       \
    if (sc) return ts == 0 ? c3_go<N, 0, false, true, true, FPW_, FC_, NHB_, DT, WD_>(grid, st, a)        \
                           : c3_go<N, 1, false, true, true, FPW_, FC_, NHB_, DT, WD_>(grid, st, a);       \
    return ts == 0 ? c3_go<N, 0, false, false, true, FPW_, FC_, NHB_, DT, WD_>(grid, st, a)               \
         : ts == 1 ? c3_go<N, 1, false, false, true, FPW_, FC_, NHB_, DT, WD_>(grid, st, a)               \
                   : c3_go<N, 2, false, false, true, FPW_, FC_, NHB_, DT, WD_>(grid, st, a);              \
  }
  if constexpr (kAllForms) { if (th4 && !w30) { if (cout == 128) C3_BODY(4, 2, 2, 2) else C3_BODY(2, 2, 2, 2) } }
  if (body == C3_BODY_M16) {   // v_mfma_f32_16x16x32 form (RD_MFMA16: rd_pack_conv3x3_m16_host made the matching weight image)
    RD_REQUIRE(wd && fold && !sc && !g1 && !ph && !s2 && sw == 1 && ts == 0 && cout == 128 && cin % 32 == 0, RD_ESHAPE,
               "conv3: the 16 x 16 x 32 form needs the 8 x 32 tile form: cout 128, folded scales, stride 1, cin a multiple of 32 (cin %d)%s", cin,
               headfuse ? " (rd_conv3x3_mfma16_ok)" : "");
    if (headfuse) return c3_go<4, 0, true, false, true, 2, 1, 2, DT, true, false, 0, true>(grid, st, a);
    return c3_go<4, 0, false, false, true, 2, 1, 2, DT, true, false, 0, true>(grid, st, a);
  }
  if (body) {   // heterogeneous tile body (c3_body): the packer made the matching weight image
    RD_REQUIRE(wd && fold && !headfuse && !g1 && !ph && sw == 1 && a.nchunk % c3_body(body).nu == 0, RD_ESHAPE,
               "conv3: tile body %d needs the 8 x 32 tile form with folded scales (%d chunks)", body, a.nchunk);
    if (body == 1) {
      RD_REQUIRE(ts == 1 && !s2, RD_ESHAPE, "conv3: body 1 is the stride-2 pair view");
      if (sc) { if (cout == 128) return c3_go<4, 1, false, true, true, 2, 1, 2, DT, true, false, 1>(grid, st, a); return c3_go<2, 1, false, true, true, 2, 1, 2, DT, true, false, 1>(grid, st, a); }
      if (cout == 128) return c3_go<4, 1, false, false, true, 2, 1, 2, DT, true, false, 1>(grid, st, a);
      return c3_go<2, 1, false, false, true, 2, 1, 2, DT, true, false, 1>(grid, st, a);
    }
    RD_REQUIRE(ts == 0 && !sc, RD_ESHAPE, "conv3: tile body %d takes all nine taps, no shortcut", body);
    if (body == 2) { if (cout == 128) return c3_go<4, 0, false, false, true, 2, 1, 2, DT, true, false, 2>(grid, st, a); return c3_go<2, 0, false, false, true, 2, 1, 2, DT, true, false, 2>(grid, st, a); }
    if (cout == 128) return c3_go<4, 0, false, false, true, 2, 1, 2, DT, true, false, 3>(grid, st, a);
    return c3_go<2, 0, false, false, true, 2, 1, 2, DT, true, false, 3>(grid, st, a);
  }
  if (ph) {
    RD_REQUIRE(wd && fold && !sc && !head && !g1 && ts == 3 && sw == 1 && ph->nph >= 1 && ph->nph <= 8, RD_ESHAPE,
               "conv3: all phases per launch need the 8 x 32 tile form (folded scales, no shortcut / output conv)");
    if (cout == 128) return c3_go<4, 3, false, false, true, 2, 1, 2, DT, true>(grid, st, a);
    return c3_go<2, 3, false, false, true, 2, 1, 2, DT, true>(grid, st, a);
  }
  if (wd && g1) {
    if (headfuse) return c3_go<4, 0, true, false, true, 2, 1, 2, DT, true, true>(grid, st, a);
    return c3_go<4, 0, false, false, true, 2, 1, 2, DT, true, true>(grid, st, a);
  }
  if (wd) {
    if (headfuse) return c3_go<4, 0, true, false, true, 2, 1, 2, DT, true>(grid, st, a);
    if (cout == 128) C3_BODY_(4, 2, 1, 2, true)
    C3_BODY_(2, 2, 1, 2, true)
  }
  if (w30) {
    if (headfuse) return c3_go<4, 0, true, false, true, 2, 1, 2, DT>(grid, st, a);
    if (cout == 128) C3_BODY(4, 2, 1, 2)
    if (hb3) C3_BODY(2, 2, 1, 3)
    if constexpr (kAllForms) C3_BODY(2, 2, 1, 2)
  }
  if (head && !sc && fold) return c3_go<4, 0, true, false, true, 4, 2, 2, DT>(grid, st, a);
  if constexpr (kAllForms) {
  if (head && !sc) return c3_go<4, 0, true, false, false, 4, 2, 2, DT>(grid, st, a);
  if (fold) { if (cout == 128) C3_BODY(4, 4, 2, 2) else C3_BODY(2, 4, 2, 2) }
#undef C3_BODY
#undef C3_BODY_
  // scale / shift applied in the epilogue (stand-alone use of the C ABI; the lowering always folds)
#define C3_PLAIN(N) return ts == 0 ? c3_go<N, 0, false, false, false, 4, 2, 2, DT>(grid, st, a) : ts == 1 ? c3_go<N, 1, false, false, false, 4, 2, 2, DT>(grid, st, a) : c3_go<N, 2, false, false, false, 4, 2, 2, DT>(grid, st, a);
  if (cout == 128) { C3_PLAIN(4) }
  C3_PLAIN(2)
#undef C3_PLAIN
  }
  return rd::fail(RD_ESHAPE, "conv3: launch form not available");
#endif
}

}  // namespace rd
