// Dev harness (not part of the product library): the experimental forms of meta16_kernel (template parameter V, k_meta.h) side by
// side with the shipping form V = 0.  tools/micro/meta_v_bench.py builds this file with hipcc, checks every form's output bit for
// bit against V = 0 and times them at the production shape.   -DMV_LIST="X(0) X(1) X(3) ..." selects the forms.
#include "../../rangedet_amd/csrc/k_meta.h"
#ifndef MV_LIST
#define MV_LIST X(0) X(1) X(3) X(5) X(7) X(8) X(9) X(11)
#endif
#ifndef MV_DT
#define MV_DT RD_BF16      // -DMV_DT=RD_F16: the fp16 instantiations (pack the weights with the same type)
#endif
using namespace rd;
extern "C" int mv_dtype(void) { return MV_DT; }
extern "C" int mv_variants(int* out, int cap) {
  int n = 0;
#define X(v) if (n < cap) out[n++] = v;
  MV_LIST
#undef X
  return n;
}
extern "C" int mv_launch(int variant, const void* data, int d_cs, int d_co, const float* coord, const void* packed, void* y,
                         int y_cs, int y_co, int B, int H, int W, void* stream) {
  MetaArgs a;
  a.data = data; a.d_cs = d_cs; a.d_co = d_co; a.coord = coord; a.packed = (const unsigned char*)packed;
  a.y = y; a.y_cs = y_cs; a.y_co = y_co; a.B = B; a.H = H; a.W = W;
  a.tiles_h = (H + 7) / 8; a.tiles_w = (W + 31) / 32; a.ntiles = B * a.tiles_h * a.tiles_w;
  a.r0 = (a.tiles_w * B) % 8 == 0 ? 8 : a.tiles_w * B;   // (MetaArgs::r0: tile order of meta16_kernel, as rd_meta_kernel_fwd sets it)
  a.m0 = meta_magic(a.r0); a.m1 = meta_magic(a.tiles_h); a.m2 = meta_magic(a.tiles_w);
  const size_t lds = meta_layout(MV_DT).wbytes + 9 * 64 * 4 * 2 + 1024 + (size_t)10 * 34 * 128 + 4096;
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int grid = a.ntiles < cus ? a.ntiles : cus;
#define X(v) if (variant == v) { allow_big_lds(meta16_kernel<8, MV_DT, v>); hipLaunchKernelGGL((meta16_kernel<8, MV_DT, v>), dim3(grid), dim3(512), lds, (hipStream_t)stream, a); return (int)hipGetLastError(); }
  MV_LIST
#undef X
  return -1;
}
