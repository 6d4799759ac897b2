// Dev harness (round 6, not part of the product library): meta16_kernel with TWELVE waves per workgroup (three per SIMD, <= 168 registers)
// and its W1 fragments read from global memory (form flag 256, k_meta.h) next to the shipping 8-wave form.  tools/micro/meta_w12_bench.py
// builds this file, checks the outputs bit for bit against the shipping form and times them at the production shape.
#include "../../rangedet_amd/csrc/k_meta.h"
using namespace rd;
template <int WAVES, int V>
static int go(const void* data, int d_cs, int d_co, const float* coord, const void* packed, void* y, int y_cs, int y_co, int B, int H, int W, void* stream) {
  MetaArgs a;
  a.data = data; a.d_cs = d_cs; a.d_co = d_co; a.coord = coord; a.packed = (const unsigned char*)packed;
  a.y = y; a.y_cs = y_cs; a.y_co = y_co; a.B = B; a.H = H; a.W = W;
  a.tiles_h = (H + WAVES - 1) / WAVES; a.tiles_w = (W + 31) / 32; a.ntiles = B * a.tiles_h * a.tiles_w;
  a.r0 = (a.tiles_w * B) % 8 == 0 ? 8 : a.tiles_w * B;
  a.m0 = meta_magic(a.r0); a.m1 = meta_magic(a.tiles_h); a.m2 = meta_magic(a.tiles_w);
  constexpr size_t W1S_B = 9 * 2 * 2 * 64 * 16, A2_B = 9 * 2 * 2 * 2 * 64 * 16;
  const size_t lds = ((V & 256) ? A2_B : W1S_B + A2_B) + 9 * 64 * 4 * 2 + 1024 + (size_t)(WAVES + 2) * 34 * 128 + ((size_t)3 * (WAVES + 2) * 34 * 4 + 255) / 256 * 256;
  if (lds > 160 * 1024) return -2;
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int grid = a.ntiles < cus ? a.ntiles : cus;
  allow_big_lds(meta16_kernel<WAVES, RD_BF16, V>);
  hipLaunchKernelGGL((meta16_kernel<WAVES, RD_BF16, V>), dim3(grid), dim3(WAVES * 64), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
// variant = 1000 * waves + form
extern "C" int mw_launch(int variant, const void* data, int d_cs, int d_co, const float* coord, const void* packed, void* y, int y_cs, int y_co,
                         int B, int H, int W, void* stream) {
#define X(WV, V) if (variant == 1000 * WV + V) return go<WV, V>(data, d_cs, d_co, coord, packed, y, y_cs, y_co, B, H, W, stream);
  X(8, 219) X(8, 475) X(12, 475) X(12, 472) X(12, 473) X(12, 408)
#undef X
  return -1;
}
extern "C" int mw_variants(int* out, int cap) {
  const int v[] = {8219, 8475, 12475, 12472, 12473, 12408};
  int n = 0;
  for (int x : v) if (n < cap) out[n++] = x;
  return n;
}
