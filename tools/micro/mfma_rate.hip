// Dev micro-benchmark: issue rate of independent v_mfma_f32_32x32x16_bf16 from 1 or 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate tools/micro/mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int FILL>
__global__ void k(float* out, long long* cyc, int iters, int zero) {
  f32x16 acc[NACC];
  s16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = zero ? 0 : (short)(0x3f80 + threadIdx.x + i); b[i] = zero ? 0 : (short)(0x3f00 + i * 3 + threadIdx.x); }
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  int x = threadIdx.x;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < FILL; ++f) { asm volatile("v_add_u32 %0, %0, 1" : "+v"(x)); }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = (float)x;
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int FILL>
void run(int threads, int zero) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 4000;
  hipLaunchKernelGGL((k<NACC, FILL>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, zero);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NACC, FILL>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, zero);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double c = 0; for (int i = 0; i < 256; ++i) c += h[i]; c /= 256;
  const double nm = (double)iters * NACC;            // MFMAs per wave
  const double flops = nm * 32768.0 * (threads / 64) * 256;
  printf("acc %2d fill %2d waves/SIMD %d zero %d: %.1f cycles per MFMA per wave, %.0f TFLOP/s, clock %.0f MHz\n", NACC, FILL, threads / 256, zero,
         c / nm, flops / (ms * 1e-3) / 1e12, c / (ms * 1e-3) / 1e6);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int zero = 1; zero >= 0; --zero) {
    run<8, 0>(256, zero); run<8, 0>(512, zero);
    run<8, 4>(256, zero); run<8, 6>(256, zero); run<8, 7>(256, zero); run<8, 8>(256, zero);
    run<16, 0>(256, zero); run<4, 0>(256, zero); run<4, 0>(512, zero);
  }
  return 0;
}
