// Dev harness (tools/micro/meta_x2_bench.py): the shipping Meta-Kernel (variant 0) and the two-fragments-per-wave form (variant 1)
// behind one launcher, in a library of its own so that a kernel edit rebuilds in seconds.
#include "../../rangedet_amd/csrc/k_meta.h"
namespace rd {
// EXPERIMENT, not in the product: measured 265 - 278 us against 231 us for meta16_kernel on the same box, with the
// compiler's schedule (AGPR-form MFMAs: +832 v_accvgpr_read per tile; with -mllvm -amdgpu-mfma-vgpr-form and MX2_FENCES=1: 44).
// ---- two pixel fragments per wave (round 3) ------------------------------------------------------------------------------
// The same arithmetic as meta16_kernel, instruction for instruction per pixel fragment (so the outputs are bit-identical), in the
// opposite corner of the occupancy trade: ONE wave per SIMD (4-wave workgroups, up to 512 registers per wave), each wave owning
// TWO rows of the 8 x 32 tile.  Every weight fragment, bias and shift vector of a tap is read from LDS once and feeds both
// fragments (36 instead of 66 ds_read_b128 per two fragments and tap), and the two fragments are two independent dependency
// chains in one instruction stream: while one's MFMA result is in flight the other's vector work issues -- what the second wave
// per SIMD did for meta16_kernel, without its second set of LDS reads.
#ifndef MX2_FENCES
#define MX2_FENCES 0
#endif
#ifndef MX2_MINBLK
#define MX2_MINBLK 1   // 2: a 256-register budget, i.e. no AGPR forms of the MFMAs (their results then need no v_accvgpr_read)
#endif
#define MX2_FENCE_TAP() { if (MX2_FENCES & 1) __builtin_amdgcn_sched_barrier(0); }
#define MX2_FENCE_MT() { if (MX2_FENCES & 2) __builtin_amdgcn_sched_barrier(0); }
#define MX2_FENCE_S2() { if (MX2_FENCES & 4) __builtin_amdgcn_sched_barrier(0); }
template <int DT = RD_BF16>
__global__ __launch_bounds__(256, MX2_MINBLK) void meta16x2_kernel(MetaArgs a) {
  using HT = H16<DT>;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  constexpr int ROWS = 8, PXB = 128, SPP = 8, HC = 34, HR = ROWS + 2, NT = 256;
  HIP_DYNAMIC_SHARED(unsigned char, smem);
  constexpr size_t W1S_B = 9 * 2 * 2 * 64 * 16, A2_B = 9 * 2 * 2 * 2 * 64 * 16, WB = W1S_B + A2_B;
  constexpr size_t CONST_B = 9 * 64 * 4 * 2 + 512 + 512;
  unsigned char* lw = smem;
  unsigned char* lc = smem + WB;
  unsigned char* halo = lc + CONST_B;
  float* chalo = (float*)(halo + HR * HC * PXB);  // [3][HR][HC]
  const float* cb1 = (const float*)lc;            // [9][64]
  const float* ct1 = cb1 + 9 * 64;                // [9][64]
  const float* cs2 = ct1 + 9 * 64 + 2 * 16 * 4;   // [64] s2, [64] t2
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int px = lane & 31, hi = lane >> 5;

  for (size_t i = tid; i < WB / 16; i += NT) ((Slot16*)lw)[i] = ((const Slot16*)a.packed)[i];
  for (size_t i = tid; i < CONST_B / 16; i += NT) ((Slot16*)lc)[i] = ((const Slot16*)(a.packed + WB))[i];
  const bf16_t* data = (const bf16_t*)a.data;
  bf16_t* yout = (bf16_t*)a.y;
  const long HW = (long)a.H * a.W;

  // halo prefetch registers (next tile's data slots / coordinates, in flight during this tile's math); the per-item halo
  // position is re-derived from the item index at fetch time (a dozen integer instructions per item and tile)
  constexpr int DITEMS = HR * HC * SPP, DU = (DITEMS + NT - 1) / NT;
  constexpr int CITEMS = 3 * HR * HC, CU = (CITEMS + NT - 1) / NT;
  Slot16 dreg[DU];
  float creg[CU];
  auto fetch = [&](int tile) {
    const int tw = tile % a.tiles_w, th = (tile / a.tiles_w) % a.tiles_h, b = tile / (a.tiles_w * a.tiles_h);
    const int h0 = th * ROWS - 1, w0 = tw * 32 - 1;
    const bf16_t* dbase = data + (size_t)b * a.H * a.W * a.d_cs + a.d_co;
    const float* cbase = a.coord + (size_t)b * 3 * HW;
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int idx = u * NT + tid, pl = idx >> 3, s = idx & 7, r = (pl * 1928) >> 16, c = pl - r * HC;   // (pl / 34 for pl < 400)
      const int ih = h0 + r, iw = w0 + c;
      dreg[u] = Slot16{0u, 0u, 0u, 0u};
      if (idx < DITEMS && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)
        dreg[u] = *(const Slot16*)(dbase + ((size_t)ih * a.W + iw) * a.d_cs + s * 8);
    }
#pragma unroll
    for (int u = 0; u < CU; ++u) {
      const int idx = u * NT + tid, ch = idx / (HR * HC), pl = idx - ch * (HR * HC), r = (pl * 1928) >> 16, c = pl - r * HC;
      const int ih = h0 + r, iw = w0 + c;
      creg[u] = 0.f;                              // im2col zero padding: outside the image the coordinate is 0
      if (idx < CITEMS && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)
        creg[u] = cbase[(size_t)ch * HW + (long)ih * a.W + iw];
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int idx = u * NT + tid, pl = idx >> 3, s = idx & 7;
      if (idx < DITEMS) *(Slot16*)(halo + pl * PXB + ((s ^ ((pl >> 1) & 7)) << 4)) = dreg[u];
    }
#pragma unroll
    for (int u = 0; u < CU; ++u)
      if (u * NT + tid < CITEMS) chalo[u * NT + tid] = creg[u];
  };

  const s16x8 w0frag = *(const s16x8*)(a.packed + meta_layout(DT).w0f + lane * 16);
  const unsigned char* w1l = lw + lane * 16;                 // + ((k*2 + mt)*2 + ks) * 1024
  const unsigned char* a2l = lw + W1S_B + lane * 16;         // + (((k*2 + ot)*2 + mt)*2 + s2) * 1024
  const float* cbl = cb1 + 16 * hi;                          // + k*64 + 32*mt + 4*q
  const float* ctl = ct1 + 16 * hi;
  const int pl0 = (2 * wv + 1) * HC + (px + 1);              // centre pixel of fragment 0 of this lane in the halo (fragment 1: + HC)
  int tile = blockIdx.x;
  if (tile < a.ntiles) fetch(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    const int tw = tile % a.tiles_w, th = (tile / a.tiles_w) % a.tiles_h, b = tile / (a.tiles_w * a.tiles_h);
    const int h0 = th * ROWS, w0 = tw * 32;
    __syncthreads();          // every wave is done with the previous tile's halos (and the weights are in place)
    commit();
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);

    float cc[2][3];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int d = 0; d < 3; ++d) cc[f][d] = chalo[d * HR * HC + pl0 + f * HC];

    f32x16 acc2[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[f][ot][r] = 0.f;

#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int dh = k / 3 - 1, dw = k % 3 - 1;
      const bool centre = k == META_CENTRE_TAP;
      MX2_FENCE_TAP();
      s16x8 hfrag[2][2] = {};
      if (!centre) {
        f32x16 pre[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const int pl = pl0 + f * HC + dh * HC + dw;
          const float r0 = chalo[pl] - cc[f][0], r1 = chalo[HR * HC + pl] - cc[f][1], r2 = chalo[2 * HR * HC + pl] - cc[f][2];
          const unsigned hxy = HT::pk(r0, r1), hz1 = HT::pk(r2, 1.0f);
          const f32x2 uxy = HT::unpk(hxy), uz1 = HT::unpk(hz1);
          const float l0 = r0 - uxy[0], l1 = r1 - uxy[1];
          const float l2 = r2 - uz1[0];
          const unsigned lxy = HT::pk(l0, l1), lz0 = HT::pk(l2, 0.f);
          unsigned pk0[4] = {hxy, hz1, hi ? 0u : lxy, hi ? 0u : lz0};
          s16x8 b0frag;
          memcpy(&b0frag, pk0, 16);
          pre[f] = HT::mfma(w0frag, b0frag, f32x16{});
        }
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            unsigned pk[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned v = HT::pk(pre[f][8 * ks + 2 * e], pre[f][8 * ks + 2 * e + 1]);
              pk[e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), (s16x2){0, 0}));
            }
            memcpy(&hfrag[f][ks], pk, 16);
          }
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        MX2_FENCE_MT();
        // bias of the 32-channel block: ONE read for both fragments (it is the C operand of their first MFMA #1)
        f32x16 bias;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bq = *(const f32x4*)(cbl + k * 64 + 32 * mt + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) bias[4 * q + e] = bq[e];
        }
        f32x16 d1[2] = {bias, bias};
        if (!centre) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const s16x8 af = *(const s16x8*)(w1l + ((k * 2 + mt) * 2 + ks) * 1024);
#pragma unroll
            for (int f = 0; f < 2; ++f) d1[f] = HT::mfma(af, hfrag[f][ks], d1[f]);
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const f32x4 t0 = *(const f32x4*)(ctl + k * 64 + 32 * mt + 8 * s2);
          const f32x4 t1v = *(const f32x4*)(ctl + k * 64 + 32 * mt + 8 * s2 + 4);
          MX2_FENCE_S2();
          s16x8 bfrag[2];
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const int pl = pl0 + f * HC + dh * HC + dw;
            const Slot16 dv = *(const Slot16*)(halo + pl * PXB + (((4 * mt + 2 * hi + s2) ^ ((pl >> 1) & 7)) << 4));
            unsigned pk[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const f32x2 x2 = HT::unpk(dv[e]);
              const f32x2 w2 = {d1[f][8 * s2 + 2 * e], d1[f][8 * s2 + 2 * e + 1]};
              const f32x2 b2 = e < 2 ? f32x2{t0[2 * e], t0[2 * e + 1]} : f32x2{t1v[2 * e - 4], t1v[2 * e - 3]};
              const f32x2 v2 = x2 * w2 + b2;
              const unsigned v = HT::pk(v2[0], v2[1]);
              pk[e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), (s16x2){0, 0}));
            }
            memcpy(&bfrag[f], pk, 16);
          }
#pragma unroll
          for (int ot = 0; ot < 2; ++ot) {
            const s16x8 af = *(const s16x8*)(a2l + (((k * 2 + ot) * 2 + mt) * 2 + s2) * 1024);
#pragma unroll
            for (int f = 0; f < 2; ++f) acc2[f][ot] = HT::mfma(af, bfrag[f], acc2[f][ot]);
          }
        }
      }
    }
    // epilogue: BN + ReLU, 16 contiguous output channels per (lane, ot)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int h = h0 + 2 * wv + f, w = w0 + px;
      if (h < a.H && w < a.W) {
        bf16_t* yp = yout + (((size_t)b * a.H + h) * a.W + w) * a.y_cs + a.y_co;
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
          const int ob = 32 * ot + 16 * hi;
          unsigned pk[8];
#pragma unroll
          for (int r = 0; r < 16; r += 2)
            pk[r >> 1] = HT::pk(fmaxf(acc2[f][ot][r] * cs2[ob + r] + cs2[64 + ob + r], 0.f),
                                fmaxf(acc2[f][ot][r + 1] * cs2[ob + r + 1] + cs2[64 + ob + r + 1], 0.f));
          *(Slot16*)(yp + ob) = Slot16{pk[0], pk[1], pk[2], pk[3]};
          *(Slot16*)(yp + ob + 8) = Slot16{pk[4], pk[5], pk[6], pk[7]};
        }
      }
    }
  }
}

}  // namespace rd
using namespace rd;
extern "C" int mx_launch(int variant, const void* data, int d_cs, int d_co, const float* coord, const void* packed, void* y,
                         int y_cs, int y_co, int B, int H, int W, void* stream) {
  MetaArgs a;
  a.data = data; a.d_cs = d_cs; a.d_co = d_co; a.coord = coord; a.packed = (const unsigned char*)packed;
  a.y = y; a.y_cs = y_cs; a.y_co = y_co; a.B = B; a.H = H; a.W = W;
  a.tiles_h = (H + 7) / 8; a.tiles_w = (W + 31) / 32; a.ntiles = B * a.tiles_h * a.tiles_w;
  a.r0 = (a.tiles_w * B) % 8 == 0 ? 8 : a.tiles_w * B;   // (MetaArgs::r0: tile order of meta16_kernel, as rd_meta_kernel_fwd sets it)
  a.m0 = meta_magic(a.r0); a.m1 = meta_magic(a.tiles_h); a.m2 = meta_magic(a.tiles_w);
  const size_t lds = meta_layout(RD_BF16).wbytes + 9 * 64 * 4 * 2 + 1024 + (size_t)10 * 34 * 128 + 4096;
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int grid = a.ntiles < cus ? a.ntiles : cus;
  if (variant == 0) {
    allow_big_lds(meta16_kernel<8, RD_BF16>);
    hipLaunchKernelGGL((meta16_kernel<8, RD_BF16>), dim3(grid), dim3(512), lds, (hipStream_t)stream, a);
  } else {
    allow_big_lds(meta16x2_kernel<RD_BF16>);
    hipLaunchKernelGGL((meta16x2_kernel<RD_BF16>), dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
  }
  return (int)hipGetLastError();
}
