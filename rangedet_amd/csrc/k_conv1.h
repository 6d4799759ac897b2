// 1x1 bf16 convolution (projection shortcuts, dla_backbone.py:34-51: stride (1,1) or (1,2)) + BN (+ReLU / residual) as a
// streaming GEMM: HBM-bound (Cin*2 bytes in, Cout*2 bytes out per output pixel), so the kernel is built around whole-line
// memory accesses.  One wave owns 32 output pixels at a time:
//   * global loads fetch whole pixel rows (a wave instruction = 64 x 16 B = consecutive 16-byte slots of 64/SP pixels);
//     a stride-2 conv simply skips the odd pixels -- their cache lines are never touched;
//   * the rows are transposed through a wave-private, XOR-swizzled LDS image into the MFMA B operand;
//   * the weight matrix lives in registers for the whole kernel (MFMA-fragment-ordered packed image, pack_taps_frag);
//   * the epilogue goes back through the same LDS image so every store instruction writes whole pixel rows.
// The next tile's rows are requested before the current tile is processed (register double buffer) when they fit.
#pragma once
#include "k_conv3.h"

namespace rd {

struct Conv1Args {
  const bf16_t* x; int x_cs, x_co;
  const unsigned char* w;
  const float* scale; const float* shift;
  const bf16_t* res; int r_cs, r_co;
  bf16_t* y; int y_cs, y_co;
  long npix;          // B * H * Wout output pixels
  int Win, Wout, sw, flags;
};

template <int SP> __device__ __forceinline__ int c1_swz(int px) {   // 16-byte-slot swizzle of pixel row px (SP slots per row)
  return SP == 2 ? (px >> 3) & 1 : SP == 8 ? (px >> 1) & 7 : SP == 4 ? (px >> 2) & 3 : px & (SP - 1);
}

template <int NCT, int NKS, bool DB>   // Cout = 32*NCT, Cin padded = 16*NKS, register double buffering of the input rows
__global__ __launch_bounds__(256) void conv1x1_stream_kernel(Conv1Args a) {
  constexpr int COUT = NCT * 32, SPI = NKS * 2, SPO = COUT / 8;        // 16-byte slots per input / output pixel row
  constexpr int PPI = 64 / SPI, NLD = 32 / PPI;                        // pixels per load instruction, loads per tile
  constexpr int RPO = 64 / SPO, NST = 32 / RPO;                        // pixels per store instruction, stores per tile
  constexpr int IMG = 32 * (SPI > SPO ? SPI : SPO) * 16;               // wave-private LDS image (input rows, then output rows)
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * IMG + 2 * COUT * 4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = lane & 31, hi = lane >> 5;
  unsigned char* img = lds + wv * IMG;
  float* Sc = (float*)(lds + 4 * IMG);
  if (tid < COUT) {
    Sc[tid] = a.scale ? a.scale[tid] : 1.f;
    Sc[COUT + tid] = a.shift ? a.shift[tid] : 0.f;
  }
  __syncthreads();
  s16x8 wf[NKS][NCT];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
    for (int j = 0; j < NCT; ++j)   // packed [32-ch chunk][tap 0][ks & 1][Cout/32][lane]
      wf[ks][j] = *(const s16x8*)(a.w + ((size_t)(ks * NCT + j) * 64 + lane) * 16);   // k-step ks = chunk 2*(ks>>1), half ks&1
  const long ntile = (a.npix + 31) / 32;
  const long wave0 = (long)blockIdx.x * 4 + wv, nwave = (long)gridDim.x * 4;
  const int lp = lane / SPI, ls = lane % SPI;                          // load role: pixel it*PPI + lp of the tile, slot ls
  auto load = [&](long tile, Slot16 (&v)[NLD]) {
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const long p = min(tile * 32 + it * PPI + lp, a.npix - 1);       // dead pixels re-read the last one, never stored
      const long row = p / a.Wout;
      const long pin = row * a.Win + (p - row * a.Wout) * a.sw;        // stride 2: even input columns only
      v[it] = *(const Slot16*)(a.x + pin * a.x_cs + a.x_co + ls * 8);
    }
  };
  const bool relu_pre = a.flags & RD_RELU_PRE, do_add = a.flags & RD_ADD, relu_post = a.flags & RD_RELU_POST;
  Slot16 cur[NLD], nxt[NLD];
  if (wave0 < ntile) load(wave0, cur);
  for (long tile = wave0; tile < ntile; tile += nwave) {
    const bool more = tile + nwave < ntile;
    if (DB && more) load(tile + nwave, nxt);
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int pr = it * PPI + lp;
      *(Slot16*)(img + pr * SPI * 16 + ((ls ^ (c1_swz<SPI>(pr) & (SPI - 1))) << 4)) = cur[it];
    }
    __builtin_amdgcn_wave_barrier();   // LDS ops of a wave are in order; this orders hipemu's lanes
    f32x16 acc[NCT];
#pragma unroll
    for (int j = 0; j < NCT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const s16x8 b = *(const s16x8*)(img + m * SPI * 16 + (((2 * ks + hi) ^ (c1_swz<SPI>(m) & (SPI - 1))) << 4));
#pragma unroll
      for (int j = 0; j < NCT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][j], b, acc[j], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    if (!DB && more) load(tile + nwave, cur);                          // single buffer: request the next rows now
    // epilogue: lane (m, hi) of acc[j] holds channels 32*j + 16*hi + r of pixel m (transposed MFMA, permuted weight rows)
    const long p = tile * 32 + m;
    const bool live = p < a.npix;
#pragma unroll
    for (int j = 0; j < NCT; ++j) {
      const int cb = j * 32 + 16 * hi;
      Slot16 rv[2] = {Slot16{0u, 0u, 0u, 0u}, Slot16{0u, 0u, 0u, 0u}};
      if (do_add && live) {
        rv[0] = *(const Slot16*)(a.res + p * a.r_cs + a.r_co + cb);
        rv[1] = *(const Slot16*)(a.res + p * a.r_cs + a.r_co + cb + 8);
      }
      unsigned pk[8];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 sc = *(const f32x4*)(Sc + cb + 4 * g4);
        const f32x4 sh = *(const f32x4*)(Sc + COUT + cb + 4 * g4);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g4 + e;
          v[e] = acc[j][r] * sc[e] + sh[e];
          if (relu_pre) v[e] = fmaxf(v[e], 0.f);
          if (do_add) {
            const unsigned w2 = rv[r >> 3][(r >> 1) & 3];
            v[e] += __uint_as_float((r & 1) ? (w2 & 0xffff0000u) : (w2 << 16));
          }
          if (relu_post) v[e] = fmaxf(v[e], 0.f);
        }
        pk[2 * g4] = f32x2_to_bf16x2(v[0], v[1]);
        pk[2 * g4 + 1] = f32x2_to_bf16x2(v[2], v[3]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
        *(Slot16*)(img + m * SPO * 16 + ((((cb >> 3) + u) ^ (c1_swz<SPO>(m) & (SPO - 1))) << 4)) =
            Slot16{pk[4 * u], pk[4 * u + 1], pk[4 * u + 2], pk[4 * u + 3]};
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NST; ++it) {
      const int pr = it * RPO + lane / SPO, sl = lane % SPO;
      const Slot16 v = *(const Slot16*)(img + pr * SPO * 16 + ((sl ^ (c1_swz<SPO>(pr) & (SPO - 1))) << 4));
      const long po = tile * 32 + pr;
      if (po < a.npix) *(Slot16*)(a.y + po * a.y_cs + a.y_co + sl * 8) = v;
    }
    __builtin_amdgcn_wave_barrier();
    if (DB && more) {
#pragma unroll
      for (int it = 0; it < NLD; ++it) cur[it] = nxt[it];
    }
  }
}

inline bool conv1_eligible(const TapList& tl, int in_stride, int out_stride, int cin, int cout, int dt, int Win, int Wq, int Wout) {
  if (dt != RD_BF16 || tl.n != 1 || tl.dh[0] != 0 || tl.dw[0] != 0 || out_stride != 1) return false;
  if ((in_stride != 1 && in_stride != 2) || Wq != (Win - 1) / in_stride + 1 || Wout != Wq) return false;
  if (cout != 64 && cout != 128) return false;
  const int nks = (cin + 15) / 16;   // 16-channel k-steps
  return (nks == 1 || nks == 4 || nks == 8) && !dev_switches().conv_v1;
}

inline int launch_conv1(const void* x, int x_cs, int x_co, const void* w, const float* scale, const float* shift,
                        const void* res, int r_cs, int r_co, void* y, int y_cs, int y_co, int B, int H, int Win, int Wout,
                        int cin, int cout, int flags, int sw, hipStream_t st) {
  const int nks = (cin + 15) / 16;
  RD_REQUIRE(x_co + nks * 16 <= x_cs, RD_ESHAPE, "conv1x1: x rows narrower than the padded cin");
  Conv1Args a;
  a.x = (const bf16_t*)x; a.x_cs = x_cs; a.x_co = x_co;
  a.w = (const unsigned char*)w; a.scale = scale; a.shift = shift;
  a.res = (const bf16_t*)res; a.r_cs = r_cs; a.r_co = r_co;
  a.y = (bf16_t*)y; a.y_cs = y_cs; a.y_co = y_co;
  a.npix = (long)B * H * Wout; a.Win = Win; a.Wout = Wout; a.sw = sw; a.flags = flags;
  const long ntile = (a.npix + 31) / 32;
  const int grid = (int)std::min<long>((ntile + 3) / 4, 8L * conv_num_cus());
  ProfScope ps(RD_PROF_CONV, st);
#define C1_CASE(N, K, D) if (cout == 32 * N && nks == K) hipLaunchKernelGGL((conv1x1_stream_kernel<N, K, D>), dim3(grid), dim3(256), 0, st, a);
  C1_CASE(2, 1, true) else C1_CASE(2, 4, true) else C1_CASE(2, 8, true) else C1_CASE(4, 1, true) else C1_CASE(4, 4, true)
  else C1_CASE(4, 8, false)
#undef C1_CASE
  return check_launch("conv1x1_stream_kernel");
}

}  // namespace rd
