// Dev probe: fragment maps of v_mfma_f32_16x16x32_{bf16,f16} on gfx950, checked against a CPU matmul (the CPU emulator of the test tier,
// tests/emu/hip/hip_runtime.h, restates the map this confirms):  A[m][k]: lane m + 16*(k/8), element k%8;  B[k][n]: lane n + 16*(k/8),
// element k%8;  D[m][n]: lane n + 16*(m/4), register m%4.
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/mfma16_probe.bin tools/micro/mfma16_probe.hip && tools/micro/mfma16_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const short* a, const short* b, float* d, int f16) {
  const int lane = threadIdx.x;
  s16x8 av, bv;
  for (int j = 0; j < 8; ++j) { av[j] = a[lane * 8 + j]; bv[j] = b[lane * 8 + j]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  if (f16) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, av), __builtin_bit_cast(h16x8, bv), c, 0, 0, 0);
  else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[lane * 4 + r] = c[r];
}
static unsigned short bf16(float v) { unsigned u; memcpy(&u, &v, 4); return (unsigned short)(u >> 16); }
static unsigned short f16(float v) { _Float16 h = (_Float16)v; unsigned short u; memcpy(&u, &h, 2); return u; }
int main() {
  for (int t = 0; t < 2; ++t) {
    float A[16][32], B[32][16];
    unsigned s = 12345u + t;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)((int)((s >> 16) % 17) - 8); };
    for (auto& r : A) for (auto& v : r) v = rnd();
    for (auto& r : B) for (auto& v : r) v = rnd();
    short ha[512], hb[512];
    for (int m = 0; m < 16; ++m) for (int kk = 0; kk < 32; ++kk) ha[(m + 16 * (kk / 8)) * 8 + kk % 8] = (short)(t ? f16(A[m][kk]) : bf16(A[m][kk]));
    for (int n = 0; n < 16; ++n) for (int kk = 0; kk < 32; ++kk) hb[(n + 16 * (kk / 8)) * 8 + kk % 8] = (short)(t ? f16(B[kk][n]) : bf16(B[kk][n]));
    short *da, *db; float* dd;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd, t);
    float hd[256]; hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
      float ref = 0; for (int kk = 0; kk < 32; ++kk) ref += A[m][kk] * B[kk][n];
      if (hd[(n + 16 * (m / 4)) * 4 + m % 4] != ref) ++bad;
    }
    printf("v_mfma_f32_16x16x32_%s: %d of 256 elements differ from the CPU matmul under the assumed fragment map\n", t ? "f16" : "bf16", bad);
  }
  return 0;
}
