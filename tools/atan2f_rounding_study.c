// how often does glibc atan2f differ from the correctly rounded value (float(atan2(double)))?
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <omp.h>
static inline uint64_t rng(uint64_t* s) { uint64_t x = *s; x ^= x << 13; x ^= x >> 7; x ^= x << 17; return *s = x; }
static inline float u01(uint64_t* s) { return (rng(s) >> 40) * (1.0f / 16777216.0f); }
int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 1000000000L;
  long diff = 0, diff2 = 0; 
#pragma omp parallel reduction(+:diff,diff2)
  {
    uint64_t s = 0x9E3779B97F4A7C15ull * (omp_get_thread_num() + 1);
    long per = n / omp_get_num_threads();
    for (long i = 0; i < per; ++i) {
      // edge of a box: corners are floats in +-100 m with edges 0.2 .. 25 m, any orientation (float differences of float coordinates)
      float cx = (u01(&s) - 0.5f) * 200.f, cy = (u01(&s) - 0.5f) * 200.f;
      float len = 0.2f + u01(&s) * 24.8f, th = (u01(&s) - 0.5f) * 6.2831853f;
      float x1 = cx, y1 = cy, x2 = cx + len * cosf(th), y2 = cy + len * sinf(th);
      float dy = y2 - y1, dx = x2 - x1;
      float a = atan2f(dy, dx);
      float b = (float)atan2((double)dy, (double)dx);
      if (a != b) { ++diff; if (fabsf(a - b) > 2.5e-7f * fmaxf(1.f, fabsf(b))) ++diff2; }
    }
  }
  printf("n=%ld glibc atan2f != float(atan2(double)): %ld (%.3e); more than ~2 ulp: %ld\n", n, diff, (double)diff / n, diff2);
  return 0;
}
