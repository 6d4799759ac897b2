// Fused Meta-Kernel unit for gfx950 -- ONE kernel for what the reference builds from ~12 MXNet ops
// (meta_kernel.py:166-240 + dla_backbone.py:92-97):
//   rel_k = coord[p+d_k] - coord[p]           (im2col zero padding: rel_k = -coord[p] outside the image)
//   h_k   = relu(W0 rel_k + b0)               3 -> 32        (VALU, K = 3 is too thin for MFMA)
//   w_k   = W1 h_k + b1                       32 -> 64       (MFMA #1)
//   a_k   = relu(s1 * (data[p+d_k] (.) w_k) + t1)            BN(576) + ReLU, s1 folded into W1/b1 per tap
//   y     = relu(s2 * (A . a) + t2)           576 -> 64      (MFMA #2, accumulated over the 9 taps)
// The 576-channel intermediates never leave registers: MFMA #1 is issued "transposed" (A operand = weights,
// B operand = per-pixel hidden vector), so its C layout (lane = pixel, regs = channels) is, after the
// element-wise product, exactly the B operand of MFMA #2 -- the channel <-> (lane,reg) permutation this needs
// is baked into the packed weights on the host (rd_pack_meta_host), giving each lane 16 CONTIGUOUS channels
// per 32-channel block (wide LDS reads / global stores).
//   workgroup = WAVES waves = WAVES rows x 32 columns, persistent over tiles; LDS: bf16 weights (108 KiB),
//   per-tap constants, and the (WAVES+2) x 34 data halo (XOR-swizzled 16-byte slots).
//   meta_kernel<RD_F32> (parity mode): v_mfma_f32_32x32x2_f32, hidden layer on the VALU, weights streamed from L2
//   (220 KiB > LDS).  meta_bf16_kernel (production, end of this file): hidden layer on the matrix cores too.
// INPUT CONTRACT: data and coordinates are FINITE.  The 16-bit production form (meta16_kernel<8, DT, 219>) applies its ReLUs as a packed
// signed 16-bit max on the converted values and splits coordinates into a high + low part: for finite inputs that is bit-identical to
// relu(x) = x > 0 ? x : 0, but a positive-signed NaN survives the integer max (the reference's relu maps NaN to 0) and an Inf / fp16
// overflow in a coordinate difference makes the low part NaN.  rd_input_transform (k_input.h) produces finite values by
// construction (missing returns are filled, ranges are clipped: rangedet/core/input.py ProcessMissValue / SepAndClipData); a caller
// that feeds its own tensors must do the same.
#pragma once
#include "rd_common.h"

namespace rd {

struct MetaLayout {  // byte offsets inside the packed parameter block
  size_t w1s, a2, b1p, t1, w0p, s2t2, w0f, total;
  size_t wbytes;  // w1s + a2 (the part held in LDS for bf16)
};
__host__ __device__ inline MetaLayout meta_layout(int dt) {
  MetaLayout L;
  const bool h = dt != RD_F32;   // RD_BF16 / RD_F16: 16-bit weights in MFMA-fragment order
  const size_t w1s = h ? 9 * 2 * 2 * 64 * 16 : 9 * 2 * 4 * 64 * 16;
  const size_t a2 = h ? 9 * 2 * 2 * 2 * 64 * 16 : 9 * 2 * 2 * 4 * 64 * 16;
  L.w1s = 0;
  L.a2 = w1s;
  L.wbytes = w1s + a2;
  L.b1p = L.wbytes;
  L.t1 = L.b1p + 9 * 64 * 4;
  L.w0p = L.t1 + 9 * 64 * 4;
  L.s2t2 = L.w0p + 2 * 16 * 4 * 4;
  L.w0f = L.s2t2 + 128 * 4;                          // bf16: A operand of the hidden-layer MFMA (64 lanes x 8 bf16)
  L.total = L.w0f + (h ? 1024 : 0);
  return L;
}
constexpr int META_CENTRE_TAP = 4;   // (dh, dw) = (0, 0)
inline int meta_perm(int blk, int m) {  // MFMA row m of 32-block blk -> channel
  return 32 * blk + 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3);
}
inline void pack_meta(const float* w0, const float* b0, const float* w1, const float* b1, const float* s1,
                      const float* t1, const float* agg, const float* s2, const float* t2, int dt, void* out) {
  const MetaLayout L = meta_layout(dt);
  unsigned char* base = (unsigned char*)out;
  memset(base, 0, L.total);
  const bool h16 = dt != RD_F32;
  auto put = [&](size_t byte_off, size_t idx, float v) {
    if (h16) ((bf16_t*)(base + byte_off))[idx] = h16_from_f32(dt, v);
    else ((float*)(base + byte_off))[idx] = v;
  };
  for (int k = 0; k < 9; ++k)
    for (int mt = 0; mt < 2; ++mt)
      for (int lane = 0; lane < 64; ++lane) {
        const int m = lane & 31, hi = lane >> 5;
        const int ch = meta_perm(mt, m);
        const float s = s1[ch * 9 + k];
        if (h16) {
          for (int ks = 0; ks < 2; ++ks)
            for (int e = 0; e < 8; ++e) {
              // hidden unit seen by B-operand element (ks, hi, e): the bf16 kernel computes the hidden layer with an
              // MFMA whose D register r = 8*ks + e of lane (px, hi) is hidden unit (r&3) + 8*(r>>2) + 4*hi
              int j = (e & 3) + 8 * (2 * ks + (e >> 2)) + 4 * hi;
              put(L.w1s, ((((size_t)k * 2 + mt) * 2 + ks) * 64 + lane) * 8 + e, s * w1[ch * 32 + j]);
            }
        } else {
          for (int q = 0; q < 16; ++q) {
            int j = 2 * q + hi;
            put(L.w1s, ((((size_t)k * 2 + mt) * 4 + (q >> 2)) * 64 + lane) * 4 + (q & 3), s * w1[ch * 32 + j]);
          }
        }
      }
  for (int k = 0; k < 9; ++k)
    for (int ot = 0; ot < 2; ++ot)
      for (int mt = 0; mt < 2; ++mt)
        for (int lane = 0; lane < 64; ++lane) {
          const int m = lane & 31, hi = lane >> 5;
          const int o = meta_perm(ot, m);
          for (int r = 0; r < 16; ++r) {
            const int c = 32 * mt + 16 * hi + r;
            const float v = agg[(size_t)o * 576 + c * 9 + k];
            if (h16)
              put(L.a2, (((((size_t)k * 2 + ot) * 2 + mt) * 2 + (r >> 3)) * 64 + lane) * 8 + (r & 7), v);
            else
              put(L.a2, (((((size_t)k * 2 + ot) * 2 + mt) * 4 + (r >> 2)) * 64 + lane) * 4 + (r & 3), v);
          }
        }
  float* fb1 = (float*)(base + L.b1p);
  float* ft1 = (float*)(base + L.t1);
  for (int k = 0; k < 9; ++k)
    for (int c = 0; c < 64; ++c) {
      // Centre tap (k = 4): rel = coord[p] - coord[p] = 0 for every pixel, so its hidden vector is relu(b0) and its dynamic
      // weight W1 relu(b0) + b1 is a per-channel CONSTANT (meta_kernel.py:193-214 evaluates the MLP on zeros there).  It is
      // folded into the tap's bias slot here (double precision) and the kernels skip both MLP layers of that tap.
      double w = b1[c];
      if (k == META_CENTRE_TAP)
        for (int j = 0; j < 32; ++j) w += (double)w1[c * 32 + j] * (double)(b0[j] > 0.f ? b0[j] : 0.f);
      fb1[k * 64 + c] = (float)((double)s1[c * 9 + k] * w);
      ft1[k * 64 + c] = t1[c * 9 + k];
    }
  float* fw0 = (float*)(base + L.w0p);
  if (h16) {
    // A operand of the ONE v_mfma_f32_32x32x16_bf16 that computes the hidden layer, row m = hidden unit m.  Weights and
    // relative coordinates are split into bf16 high + low parts (v = vh + vl, vh = bf16(v), vl = bf16(v - vh)) and the
    // 16 K slots carry the three significant partial products -- (Wh + Wl)(xh + xl) + b up to the Wl*xl term (2^-18):
    //   B (per pixel):  k 0..3 = xh yh zh 1 | k 4..7 = xl yl zl 0 | k 8..11 = xh yh zh 1 | k 12..15 = 0
    //   A (per unit m): k 0..3 = Wh[0..2] bh | k 4..7 = Wh[0..2] 0 | k 8..11 = Wl[0..2] bl | k 12..15 = 0
    // lane (m, hi) holds k = 8*hi .. 8*hi+7.
    bf16_t* fa = (bf16_t*)(base + L.w0f);
    for (int lane = 0; lane < 64; ++lane) {
      const int m = lane & 31, hi = lane >> 5;
      float v[4] = {w0[m * 3], w0[m * 3 + 1], w0[m * 3 + 2], b0[m]};
      for (int e = 0; e < 4; ++e) {
        const bf16_t h = h16_from_f32(dt, v[e]);
        const bf16_t l = h16_from_f32(dt, v[e] - h16_to_f32(dt, h));
        fa[lane * 8 + e] = hi ? l : h;                               // k 0..3 : high parts | k 8..11: low parts
        fa[lane * 8 + 4 + e] = (hi || e == 3) ? (bf16_t)0 : h;      // k 4..7 : high weights against the low coordinates
      }
    }
  } else
  for (int hi = 0; hi < 2; ++hi)
    for (int i = 0; i < 16; ++i) {
      int j = 2 * i + hi;
      fw0[(hi * 16 + i) * 4 + 0] = w0[j * 3 + 0];
      fw0[(hi * 16 + i) * 4 + 1] = w0[j * 3 + 1];
      fw0[(hi * 16 + i) * 4 + 2] = w0[j * 3 + 2];
      fw0[(hi * 16 + i) * 4 + 3] = b0[j];
    }
  float* fs = (float*)(base + L.s2t2);
  for (int o = 0; o < 64; ++o) {
    fs[o] = s2[o];
    fs[64 + o] = t2[o];
  }
}

struct MetaArgs {
  const void* data; int d_cs, d_co;
  const float* coord;  // NCHW (B,3,H,W)
  const unsigned char* packed;
  void* y; int y_cs, y_co;
  int B, H, W;
  int tiles_h, tiles_w, ntiles;
  // Tile order of meta16_kernel: list position v -> (x = v % r0, j = v / r0) -> row tile j % tiles_h of column strip
  // (j / tiles_h) * r0 + x, where strip = column tile + tiles_w * image.  r0 = 8 is the XCD-aware order (same idea as k_conv3.h
  // Conv3Args::xcd): workgroup v runs on XCD v % 8, which walks its own strips top to bottom, so the halo rows two vertically
  // adjacent tiles share are read through ONE XCD's L2 (a pure permutation of the tile list when tiles_w * B divides by 8);
  // r0 = tiles_w * B is the plain order (strip fastest).  m0 / m1 / m2 = meta_magic of r0 / tiles_h / tiles_w: the three divisions
  // of the decode are multiply-high operations.
  int r0; unsigned m0, m1, m2;
};
// floor(n / d) = umulhi(n, ceil(2^32 / d)) for n * d < 2^32 (the error term n * (m*d - 2^32) stays below 2^32); d = 1 -> 0 = "n itself"
inline unsigned meta_magic(int d) { return d == 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }
__device__ __forceinline__ int meta_div(int n, unsigned m) { return m ? (int)(((unsigned long long)(unsigned)n * m) >> 32) : n; }
// list position -> (column tile, row tile, image)
__device__ __forceinline__ void meta_tile(const MetaArgs& a, int v, int& tw, int& th, int& b) {
  const int j = meta_div(v, a.m0), x = v - j * a.r0;
  const int q = meta_div(j, a.m1);
  th = j - q * a.tiles_h;
  const int strip = q * a.r0 + x;
  b = meta_div(strip, a.m2);
  tw = strip - b * a.tiles_w;
  // The values are uniform, but the address arithmetic they feed must stay in the vector unit: with scalar tile coordinates hipcc
  // splits every per-lane address into a scalar part and a loop-invariant vector part, hoists the vector parts out of the tile
  // loop and runs the nine taps with 16 registers less (256 + spills instead of 240; measured 277 instead of 263 us).
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(tw), "+v"(th), "+v"(b));
#endif
}

template <int DT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void meta_kernel(MetaArgs a) {
  static_assert(DT == RD_F32, "the 16-bit types run meta16_kernel (its packed W0 / W1 layouts differ); this is the fp32 parity kernel");
  using E = Elem<DT>;
  using T = typename E::T;
  constexpr int PXB = 256;             // bytes per pixel (64 channels)
  constexpr int SPP = PXB / 16;        // 16-byte slots per pixel
  constexpr int HC = 34;               // halo columns
  HIP_DYNAMIC_SHARED(unsigned char, smem);
  // LDS carve: [consts] [halo]; the weights (220 KiB) stream from L2
  constexpr size_t W1S_B = 9 * 2 * 4 * 64 * 16;
  constexpr size_t A2_B = 9 * 2 * 2 * 4 * 64 * 16;
  constexpr size_t WB = W1S_B + A2_B;
  constexpr size_t CONST_B = 9 * 64 * 4 * 2 + 512 + 512;
  unsigned char* lc = smem;                       // consts
  unsigned char* halo = lc + CONST_B;
  const float* cb1 = (const float*)lc;            // [9][64]
  const float* ct1 = cb1 + 9 * 64;                // [9][64]
  const float* cw0 = ct1 + 9 * 64;                // [2][16][4]
  const float* cs2 = cw0 + 2 * 16 * 4;            // [64] s2, [64] t2
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int px = lane & 31, hi = lane >> 5;
  const int NT = WAVES * 64;

  for (size_t i = tid; i < CONST_B / 16; i += NT) ((Slot16*)lc)[i] = ((const Slot16*)(a.packed + WB))[i];
  const unsigned char* w1s = a.packed;
  const unsigned char* a2w = w1s + W1S_B;

  const T* data = (const T*)a.data;
  T* yout = (T*)a.y;
  const long HW = (long)a.H * a.W;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int tw = tile % a.tiles_w;
    const int th = (tile / a.tiles_w) % a.tiles_h;
    const int b = tile / (a.tiles_w * a.tiles_h);
    const int h0 = th * WAVES, w0 = tw * 32;
    __syncthreads();
    for (int idx = tid; idx < (WAVES + 2) * HC * SPP; idx += NT) {
      const int pl = idx / SPP, s = idx - pl * SPP;
      const int r = pl / HC, c = pl - r * HC;
      const int ih = h0 - 1 + r, iw = w0 - 1 + c;
      Slot16 v = {0u, 0u, 0u, 0u};
      if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W)
        v = *(const Slot16*)(data + (((size_t)b * a.H + ih) * a.W + iw) * a.d_cs + a.d_co + s * E::CH);
      *(Slot16*)(halo + pl * PXB + ((s ^ (pl & 15)) << 4)) = v;
    }
    __syncthreads();

    const int h = h0 + wv, w = w0 + px;
    const bool live = (h < a.H) && (w < a.W);
    const int wc = min(w, a.W - 1), hc = min(h, a.H - 1);
    const float* cbase = a.coord + (size_t)b * 3 * HW;
    const float c0 = cbase[0 * HW + (long)hc * a.W + wc];
    const float c1 = cbase[1 * HW + (long)hc * a.W + wc];
    const float c2 = cbase[2 * HW + (long)hc * a.W + wc];

    f32x16 acc2[2];
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[ot][r] = 0.f;

    for (int k = 0; k < 9; ++k) {
      const int dh = k / 3 - 1, dw = k % 3 - 1;
      const int nh = hc + dh, nw = wc + dw;
      float r0 = -c0, r1 = -c1, r2 = -c2;
      if (nh >= 0 && nh < a.H && nw >= 0 && nw < a.W) {
        r0 = cbase[0 * HW + (long)nh * a.W + nw] - c0;
        r1 = cbase[1 * HW + (long)nh * a.W + nw] - c1;
        r2 = cbase[2 * HW + (long)nh * a.W + nw] - c2;
      }
      float hv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const f32x4 wq = *(const f32x4*)(cw0 + (hi * 16 + i) * 4);
        hv[i] = fmaxf(wq[0] * r0 + wq[1] * r1 + wq[2] * r2 + wq[3], 0.f);
      }
      // MFMA #1: D1[ch][px] = (s1 W1)[ch][:] . h[:]  + s1*b1
      f32x16 d1[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bq = *(const f32x4*)(cb1 + k * 64 + 32 * mt + 16 * hi + 4 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) d1[mt][4 * g + e] = bq[e];
        }
      if (k != META_CENTRE_TAP)   // (centre tap: the dynamic weight is the constant already in d1, see pack_meta)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const f32x4 af = *(const f32x4*)(w1s + ((((size_t)k * 2 + mt) * 4 + q4) * 64 + lane) * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            d1[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], hv[4 * q4 + e], d1[mt], 0, 0, 0);
        }
      // element-wise: a = relu(data[p+d] * d1 + t1), channels 32mt+16hi+r of the neighbour pixel (from the halo)
      const int pl = (wv + 1 + dh) * HC + (px + 1 + dw);
      const unsigned char* hp = halo + pl * PXB;
      const int swz = pl & 15;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float av[16];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const f32x4 dv = *(const f32x4*)(hp + (((8 * mt + 4 * hi + s4) ^ swz) << 4));
#pragma unroll
          for (int e = 0; e < 4; ++e) av[4 * s4 + e] = dv[e];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 tq = *(const f32x4*)(ct1 + k * 64 + 32 * mt + 16 * hi + 4 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) av[4 * g + e] = fmaxf(av[4 * g + e] * d1[mt][4 * g + e] + tq[e], 0.f);
        }
        // MFMA #2: acc2[o][px] += A[o][(ch,k)] . a[ch]
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
          for (int ot = 0; ot < 2; ++ot) {
            const f32x4 af = *(const f32x4*)(a2w + (((((size_t)k * 2 + ot) * 2 + mt) * 4 + r4) * 64 + lane) * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc2[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], av[4 * r4 + e], acc2[ot], 0, 0, 0);
          }
      }
    }
    // epilogue: BN + ReLU, 16 contiguous output channels per (lane, ot)
    if (live) {
      T* yp = yout + (((size_t)b * a.H + h) * a.W + w) * a.y_cs + a.y_co;
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) {
        const int ob = 32 * ot + 16 * hi;
        T tmp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tmp[r] = E::from_f32(fmaxf(acc2[ot][r] * cs2[ob + r] + cs2[64 + ob + r], 0.f));
        Slot16 pk[sizeof(T)];  // 16 elements = sizeof(T) slots of 16 bytes
        memcpy(pk, tmp, sizeof(tmp));
#pragma unroll
        for (int i = 0; i < (int)sizeof(T); ++i) *(Slot16*)(yp + ob + i * E::CH) = pk[i];
      }
    }
  }
}


// ---- bf16 production kernel ----------------------------------------------------------------------------------------
// Same math as meta_kernel<RD_BF16>, restructured so that the vector ALU is no longer the bottleneck:
//   * the 3 -> 32 hidden layer runs on the matrix cores as ONE bf16 MFMA on high/low split operands (error 2^-18, see
//     pack_meta; the two v_mfma_f32_32x32x2_f32 it replaces cost 128 of the ~510 MFMA cycles of a tap);
//     its D layout IS the B operand of MFMA #1 (the hidden-unit permutation is baked into the packed W1);
//   * the element-wise stage uses packed fp32 FMAs, packed bf16 conversion and an integer packed max as ReLU;
//   * point coordinates come from an LDS halo (one coalesced fetch per tile instead of 27 global loads per pixel);
//   * the next tile's data / coordinate halo is fetched into registers while the current tile is computed.
// LDS: weights 108 KiB + constants 5.5 KiB + data halo 42.5 KiB + coordinate halo 4 KiB = 160 KiB, one workgroup per CU.
// DT = RD_BF16 or RD_F16 (same structure; the high / low split of MFMA #0 then carries 2^-18 resp. 2^-24 relative error).
// V: form of the tap loop (bit flags; every combination computes bit-identical results -- tools/micro/meta_v_bench.py checks that on
// the GPU and times them side by side, profiles/EXPERIMENTS.md round 4).  V = 0 is the plain loop of rounds 2 - 4a:
//   1   explicit software pipeline: the two MFMA #1 of (tap, block) step n + 1 are issued before the element-wise stage of step n
//   2   ... and the hidden layer (MFMA #0 + ReLU + pack) of tap k + 2 during block 1 of tap k
//   4   ... with a scheduling fence (vector / matrix instructions may not cross) between the look-ahead group and the current step
//   8   halo fetch addresses as 32-bit offsets from 24-bit multiplies (no 64-bit / quarter-rate integer multiplies per tile)
//   16  both fragments of a block are converted first, then its four MFMA #2 are issued together
//   32  MFMA #1 two steps ahead (three dynamic-weight buffers)
//   64  the epilogue's ReLU as a packed 16-bit integer max after the conversion (16 instead of 32 instructions)
//   128 hidden-layer operand without the two per-tap selects of the upper lane half (its k-slots 12 .. 15 meet zero weights)
// (Forms measured and removed again -- the commit "Meta-Kernel: tap loop as an explicit software pipeline" has them: 256 / 512 timing
//  ablations without the per-tap constant reads / with two vector instructions per pair; 1024 the neighbour's 16-bit channels converted
//  to fp32 by an MFMA with a 0 / 1 permutation matrix, exact and bit-identical but 240 us.)
// META_FORM = 219 (1 + 2 + 8 + 16 + 64 + 128) is what rd_meta_kernel_fwd launches: 210 - 215 us against 225 - 227 us for V = 0 on the same
// boxes (gpurun_out/r4m4, r4m5).  What did NOT pay: the fence (4), two steps of look-ahead (32), the conversion MFMA (1024: 240 us --
// every added MFMA sits in a step's dependent chain and costs ~80 wave cycles, 2.5 x its pipe time).
constexpr int META_FORM = 219;
template <int WAVES, int DT = RD_BF16, int V = META_FORM>
__global__ __launch_bounds__(WAVES * 64) void meta16_kernel(MetaArgs a) {
  static_assert((V & ~0x1FF) == 0, "meta16_kernel: unknown form flag");
  // 256 (round 6 experiment, harness only -- tools/micro/meta_w12.hip): the 36 KB of W1 fragments are read from global memory (L2 / L1
  // resident, every wave of the chip reads the same 4 KB per tap) instead of LDS, which frees the LDS a 12-wave workgroup needs for its
  // 14-row halo: three waves per SIMD at <= 168 registers instead of two ("the W1-ring + third-wave form")
  constexpr bool W1G = (V & 256) != 0;
  using HT = H16<DT>;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  constexpr int PXB = 128, SPP = 8, HC = 34, HR = WAVES + 2, NT = WAVES * 64;
  HIP_DYNAMIC_SHARED(unsigned char, smem);
  constexpr size_t W1S_B = 9 * 2 * 2 * 64 * 16, A2_B = 9 * 2 * 2 * 2 * 64 * 16, WB = W1S_B + A2_B;
  constexpr size_t CONST_B = 9 * 64 * 4 * 2 + 512 + 512;
  constexpr size_t WBL = W1G ? A2_B : WB;         // weight bytes held in LDS
  unsigned char* lw = smem;
  unsigned char* lc = smem + WBL;
  unsigned char* halo = lc + CONST_B;
  float* chalo = (float*)(halo + HR * HC * PXB);  // [3][HR][HC]
  const float* cb1 = (const float*)lc;            // [9][64]
  const float* ct1 = cb1 + 9 * 64;                // [9][64]
  const float* cs2 = ct1 + 9 * 64 + 2 * 16 * 4;   // [64] s2, [64] t2
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int px = lane & 31, hi = lane >> 5;

  for (size_t i = tid; i < WBL / 16; i += NT) ((Slot16*)lw)[i] = ((const Slot16*)(a.packed + (W1G ? W1S_B : 0)))[i];
  for (size_t i = tid; i < CONST_B / 16; i += NT) ((Slot16*)lc)[i] = ((const Slot16*)(a.packed + WB))[i];
  const bf16_t* data = (const bf16_t*)a.data;
  bf16_t* yout = (bf16_t*)a.y;
  const long HW = (long)a.H * a.W;

  // halo prefetch registers: data slots idx = u*NT + tid (u < DU), coordinate floats idx = u*NT + tid (u < CU).  Everything
  // that only depends on the thread (halo row / column / slot, LDS addresses) is computed ONCE here; per tile only the image
  // position and the bounds test remain (this kernel is instruction-issue bound: 70 % of the SIMD issue slots are taken,
  // profiles/r02_meta_pmc.txt, so per-tile integer work counts)
  constexpr int DITEMS = HR * HC * SPP, DU = (DITEMS + NT - 1) / NT;
  constexpr int CITEMS = 3 * HR * HC, CU = (CITEMS + NT - 1) / NT;
  Slot16 dreg[DU];
  float creg[CU];
  short drow[DU], dcol[DU], crow[CU], ccol[CU];
  int dlds[DU], dsrc[DU];                          // LDS byte address of the slot; element offset of its 8 channels
  int cch[CU];
#pragma unroll
  for (int u = 0; u < DU; ++u) {
    const int idx = u * NT + tid, pl = idx / SPP, s = idx - pl * SPP;
    drow[u] = (short)(pl / HC); dcol[u] = (short)(pl - (pl / HC) * HC);
    dlds[u] = idx < DITEMS ? pl * PXB + ((s ^ ((pl >> 1) & 7)) << 4) : -1;
    dsrc[u] = a.d_co + s * 8;
  }
#pragma unroll
  for (int u = 0; u < CU; ++u) {
    const int idx = u * NT + tid, ch = idx / (HR * HC), pl = idx - ch * (HR * HC);
    crow[u] = (short)(pl / HC); ccol[u] = (short)(pl - (pl / HC) * HC);
    cch[u] = idx < CITEMS ? ch : -1;
  }
  auto fetch = [&](int tile) {
    int tw, th, b;
    meta_tile(a, tile, tw, th, b);
    const int h0 = th * WAVES - 1, w0 = tw * 32 - 1;
    const bf16_t* dbase = data + (size_t)b * a.H * a.W * a.d_cs;
    const float* cbase = a.coord + (size_t)b * 3 * HW;
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int ih = h0 + drow[u], iw = w0 + dcol[u];
      dreg[u] = Slot16{0u, 0u, 0u, 0u};
      if (dlds[u] >= 0 && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W) {
        if constexpr (V & 8) dreg[u] = *(const Slot16*)(dbase + (size_t)(unsigned)(__mul24(__mul24(ih, a.W) + iw, a.d_cs) + dsrc[u]));
        else dreg[u] = *(const Slot16*)(dbase + ((size_t)ih * a.W + iw) * a.d_cs + dsrc[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < CU; ++u) {
      const int ih = h0 + crow[u], iw = w0 + ccol[u];
      creg[u] = 0.f;                              // im2col zero padding: outside the image the coordinate is 0
      if (cch[u] >= 0 && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W) {
        if constexpr (V & 8) creg[u] = cbase[(size_t)(unsigned)(__mul24(cch[u], (int)HW) + __mul24(ih, a.W) + iw)];
        else creg[u] = cbase[(size_t)cch[u] * HW + (long)ih * a.W + iw];
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < DU; ++u)
      if (dlds[u] >= 0) *(Slot16*)(halo + dlds[u]) = dreg[u];
#pragma unroll
    for (int u = 0; u < CU; ++u)
      if (cch[u] >= 0) chalo[u * NT + tid] = creg[u];
  };

  // A operand of the hidden-layer MFMA (see pack_meta): four registers for the whole kernel
  const s16x8 w0frag = *(const s16x8*)(a.packed + meta_layout(DT).w0f + lane * 16);
  // per-lane LDS addresses that do not depend on the tile: everything a tap adds to them is a compile-time constant
  const unsigned char* w1l;                                  // + ((k*2 + mt)*2 + ks) * 1024
  if constexpr (W1G) w1l = a.packed + lane * 16; else w1l = lw + lane * 16;
  const unsigned char* a2l = lw + (W1G ? 0 : W1S_B) + lane * 16;   // + (((k*2 + ot)*2 + mt)*2 + s2) * 1024
  const float* cbl = cb1 + 16 * hi;                          // + k*64 + 32*mt + 4*q
  const float* ctl = ct1 + 16 * hi;
  const int pl0 = (wv + 1) * HC + (px + 1);                  // centre pixel of this lane in the halo
  int tile = blockIdx.x;
  if (tile < a.ntiles) fetch(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    int tw, th, b;
    meta_tile(a, tile, tw, th, b);
    const int h0 = th * WAVES, w0 = tw * 32;
    __syncthreads();          // every wave is done with the previous tile's halos (and the weights are in place)
    commit();
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);   // in flight during this tile's math
    // (issuing it at tap 3 instead -- 214 instead of 240 registers -- changes nothing: 272 vs 272 us, gpurun_out/r3zg)

    const int h = h0 + wv, w = w0 + px;
    const bool live = (h < a.H) && (w < a.W);
    const float c0 = chalo[pl0], c1 = chalo[HR * HC + pl0], c2 = chalo[2 * HR * HC + pl0];

    f32x16 acc2[2];
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[ot][r] = 0.f;

    if constexpr (V & 1) {
      // the same arithmetic as the loop below (bit-identical results), as three stages of an explicit software pipeline
      auto hidden = [&](int k, s16x8 (&hf)[2]) {
        const int pl = pl0 + (k / 3 - 1) * HC + (k % 3 - 1);
        const float r0 = chalo[pl] - c0, r1 = chalo[HR * HC + pl] - c1, r2 = chalo[2 * HR * HC + pl] - c2;
        const unsigned hxy = HT::pk(r0, r1), hz1 = HT::pk(r2, 1.0f);
        const f32x2 uxy = HT::unpk(hxy), uz1 = HT::unpk(hz1);
        const float l0 = r0 - uxy[0], l1 = r1 - uxy[1];
        const float l2 = r2 - uz1[0];
        const unsigned lxy = HT::pk(l0, l1), lz0 = HT::pk(l2, 0.f);
        // (128: no selects -- k-slots 12 .. 15 of the upper lane half meet zero weights in the A operand, pack_meta, so what they hold
        //  does not matter as long as it is finite, and the low parts are)
        unsigned pk0[4] = {hxy, hz1, (V & 128) || !hi ? lxy : 0u, (V & 128) || !hi ? lz0 : 0u};
        s16x8 b0frag;
        memcpy(&b0frag, pk0, 16);
        const f32x16 pre = HT::mfma(w0frag, b0frag, f32x16{});
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          unsigned pk[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned v = HT::pk(pre[8 * ks + 2 * e], pre[8 * ks + 2 * e + 1]);
            pk[e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), (s16x2){0, 0}));
          }
          memcpy(&hf[ks], pk, 16);
        }
      };
      auto d1calc = [&](int k, int mt, const s16x8 (&hf)[2]) {
        f32x16 d1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bq = *(const f32x4*)(cbl + k * 64 + 32 * mt + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) d1[4 * q + e] = bq[e];
        }
        if (k != META_CENTRE_TAP) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const s16x8 af = *(const s16x8*)(w1l + ((k * 2 + mt) * 2 + ks) * 1024);
            d1 = HT::mfma(af, hf[ks], d1);
          }
        }
        return d1;
      };
      auto elem = [&](int k, int mt, const f32x16& d1) {
        const int pl = pl0 + (k / 3 - 1) * HC + (k % 3 - 1);
        const unsigned char* hp = halo + pl * PXB;
        const int swz = (pl >> 1) & 7;
        s16x8 bfr[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const Slot16 dv = *(const Slot16*)(hp + (((4 * mt + 2 * hi + s2) ^ swz) << 4));
          const f32x4 t0 = *(const f32x4*)(ctl + k * 64 + 32 * mt + 8 * s2);
          const f32x4 t1v = *(const f32x4*)(ctl + k * 64 + 32 * mt + 8 * s2 + 4);
          unsigned pk[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x2 x2 = HT::unpk(dv[e]);
            const f32x2 w2 = {d1[8 * s2 + 2 * e], d1[8 * s2 + 2 * e + 1]};
            const f32x2 b2 = e < 2 ? f32x2{t0[2 * e], t0[2 * e + 1]} : f32x2{t1v[2 * e - 4], t1v[2 * e - 3]};
            const f32x2 v2 = x2 * w2 + b2;
            const unsigned v = HT::pk(v2[0], v2[1]);
            pk[e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), (s16x2){0, 0}));
          }
          memcpy(&bfr[s2], pk, 16);
          if constexpr (!(V & 16)) {
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
              const s16x8 af = *(const s16x8*)(a2l + (((k * 2 + ot) * 2 + mt) * 2 + s2) * 1024);
              acc2[ot] = HT::mfma(af, bfr[s2], acc2[ot]);
            }
          }
        }
        if constexpr (V & 16) {   // both fragments of the block first, then its four MFMA #2
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
              const s16x8 af = *(const s16x8*)(a2l + (((k * 2 + ot) * 2 + mt) * 2 + s2) * 1024);
              acc2[ot] = HT::mfma(af, bfr[s2], acc2[ot]);
            }
        }
      };
      s16x8 hf[2][2] = {};
      if constexpr (V & 32) {
        // MFMA #1 TWO steps ahead (three dynamic-weight buffers), the hidden layer a whole tap ahead of that
        f32x16 d1c[3];
        hidden(0, hf[0]);
        hidden(1, hf[1]);
        d1c[0] = d1calc(0, 0, hf[0]);
        d1c[1] = d1calc(0, 1, hf[0]);
#pragma unroll
        for (int st = 0; st < 18; ++st) {
          const int k = st >> 1, mt = st & 1;
          if (st + 2 < 18) {
            const int k2 = (st + 2) >> 1;       // = k + 1, block mt
            d1c[(st + 2) % 3] = d1calc(k2, mt, hf[k2 & 1]);
          }
          elem(k, mt, d1c[st % 3]);
          // hf[k & 1] was last read by d1calc(k, 1), issued at step (k - 1, 1): free from step (k, 0) on; tap k + 2 goes in at (k, 1)
          if (mt == 1 && k + 2 < 9 && k + 2 != META_CENTRE_TAP) hidden(k + 2, hf[k & 1]);
        }
      } else {
      f32x16 d1b[2];
      hidden(0, hf[0]);
      if constexpr (V & 2) hidden(1, hf[1]);
      d1b[0] = d1calc(0, 0, hf[0]);
#pragma unroll
      for (int st = 0; st < 18; ++st) {
        const int k = st >> 1, mt = st & 1;
        if (st + 1 < 18) {
          const int k2 = (st + 1) >> 1, mt2 = (st + 1) & 1;
          if constexpr (V & 2) {   // hidden layer two steps ahead: tap k + 2's after the last use of the buffer (tap k, block 1)
          } else if (mt2 == 0 && k2 != META_CENTRE_TAP) hidden(k2, hf[k2 & 1]);
          d1b[(st + 1) & 1] = d1calc(k2, mt2, hf[k2 & 1]);
          if constexpr (V & 4) __builtin_amdgcn_sched_barrier(0x94);   // scalar / memory instructions may cross, vector / matrix ones may not
        }
        elem(k, mt, d1b[st & 1]);
        if constexpr (V & 2) {
          // buffer hf[k & 1] is free once d1calc(k, 1) has been issued, i.e. after step (k, 0): tap k + 2 goes in during step (k, 1)
          if (mt == 1 && k + 2 < 9 && k + 2 != META_CENTRE_TAP) hidden(k + 2, hf[k & 1]);
        }
      }
      }
    } else {
    // the nine taps, fully unrolled: tap offsets, weight / constant addresses are immediates of the LDS instructions
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      constexpr int dummy = 0; (void)dummy;
      const int dh = k / 3 - 1, dw = k % 3 - 1;
      const int pl = pl0 + dh * HC + dw;
      // (centre tap: rel = 0 for every pixel, the dynamic weight is the per-channel constant pack_meta put into the bias slot:
      //  no hidden layer, no MFMA #1 -- 5 of the 117 MFMAs and ~30 of the ~1000 vector instructions of a pixel fragment)
      const bool centre = k == META_CENTRE_TAP;
      f32x16 pre = f32x16{};
      s16x8 hfrag[2] = {};
      if (!centre) {
      const float r0 = chalo[pl] - c0, r1 = chalo[HR * HC + pl] - c1, r2 = chalo[2 * HR * HC + pl] - c2;
      // MFMA #0: pre[j][px] = W0[j][0..2] . rel + b0[j], one bf16 MFMA on high / low split operands (~fp32 accurate)
      {
        const unsigned hxy = HT::pk(r0, r1), hz1 = HT::pk(r2, 1.0f);
        const f32x2 uxy = HT::unpk(hxy), uz1 = HT::unpk(hz1);
        const float l0 = r0 - uxy[0], l1 = r1 - uxy[1];
        const float l2 = r2 - uz1[0];
        const unsigned lxy = HT::pk(l0, l1), lz0 = HT::pk(l2, 0.f);
        unsigned pk0[4] = {hxy, hz1, hi ? 0u : lxy, hi ? 0u : lz0};
        s16x8 b0frag;
        memcpy(&b0frag, pk0, 16);
        pre = HT::mfma(w0frag, b0frag, f32x16{});
      }
      // hidden vector as the two B fragments of MFMA #1 (ReLU on the packed pairs: negative bf16 = negative int16)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        unsigned pk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned v = HT::pk(pre[8 * ks + 2 * e], pre[8 * ks + 2 * e + 1]);
          pk[e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), (s16x2){0, 0}));
        }
        memcpy(&hfrag[ks], pk, 16);
      }
      }
      const unsigned char* hp = halo + pl * PXB;
      const int swz = (pl >> 1) & 7;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        // MFMA #1: D1[ch][px] = (s1 W1)[ch][:] . relu(pre)[:] + s1*b1, one 32-channel block at a time
        f32x16 d1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bq = *(const f32x4*)(cbl + k * 64 + 32 * mt + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) d1[4 * q + e] = bq[e];
        }
        if (!centre) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const s16x8 af = *(const s16x8*)(w1l + ((k * 2 + mt) * 2 + ks) * 1024);
          d1 = HT::mfma(af, hfrag[ks], d1);
        }
        }
        // element-wise: a = relu(data[p+d] * d1 + t1), channels 32mt+16hi+r of the neighbour pixel (from the halo)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const Slot16 dv = *(const Slot16*)(hp + (((4 * mt + 2 * hi + s2) ^ swz) << 4));
          const f32x4 t0 = *(const f32x4*)(ctl + k * 64 + 32 * mt + 8 * s2);
          const f32x4 t1v = *(const f32x4*)(ctl + k * 64 + 32 * mt + 8 * s2 + 4);
          unsigned pk[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x2 x2 = HT::unpk(dv[e]);
            const f32x2 w2 = {d1[8 * s2 + 2 * e], d1[8 * s2 + 2 * e + 1]};
            const f32x2 b2 = e < 2 ? f32x2{t0[2 * e], t0[2 * e + 1]} : f32x2{t1v[2 * e - 4], t1v[2 * e - 3]};
            const f32x2 v2 = x2 * w2 + b2;
            const unsigned v = HT::pk(v2[0], v2[1]);
            pk[e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), (s16x2){0, 0}));
          }
          s16x8 bfrag;
          memcpy(&bfrag, pk, 16);
          // MFMA #2: acc2[o][px] += A[o][(ch,k)] . a[ch]
#pragma unroll
          for (int ot = 0; ot < 2; ++ot) {
            const s16x8 af = *(const s16x8*)(a2l + (((k * 2 + ot) * 2 + mt) * 2 + s2) * 1024);
            acc2[ot] = HT::mfma(af, bfrag, acc2[ot]);
          }
        }
      }
    }
    }
    // epilogue: BN + ReLU, 16 contiguous output channels per (lane, ot)
    if (live) {
      bf16_t* yp = yout + (((size_t)b * a.H + h) * a.W + w) * a.y_cs + a.y_co;
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) {
        const int ob = 32 * ot + 16 * hi;
        unsigned pk[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          if constexpr (V & 64) {   // ReLU after the 16-bit conversion, as a packed integer max (same bits: a negative value converts to a negative pattern)
            const unsigned v = HT::pk(acc2[ot][r] * cs2[ob + r] + cs2[64 + ob + r], acc2[ot][r + 1] * cs2[ob + r + 1] + cs2[64 + ob + r + 1]);
            pk[r >> 1] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), (s16x2){0, 0}));
          } else
          pk[r >> 1] = HT::pk(fmaxf(acc2[ot][r] * cs2[ob + r] + cs2[64 + ob + r], 0.f),
                                       fmaxf(acc2[ot][r + 1] * cs2[ob + r + 1] + cs2[64 + ob + r + 1], 0.f));
        }
        *(Slot16*)(yp + ob) = Slot16{pk[0], pk[1], pk[2], pk[3]};
        *(Slot16*)(yp + ob + 8) = Slot16{pk[4], pk[5], pk[6], pk[7]};
      }
    }
  }
}

}  // namespace rd
