// restatement of the fdlibm float atan / atan2 that glibc 2.35 ships (sysdeps/ieee754/flt-32/s_atanf.c, e_atan2f.c), checked bitwise
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>
#pragma STDC FP_CONTRACT OFF
static inline uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bitsf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static const float atanhi[] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
static const float atanlo[] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
static const float aT[] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                           6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
static float r_atanf(float x) {
  float w, s1, s2, z;
  int32_t ix, hx, id;
  hx = (int32_t)fbits(x);
  ix = hx & 0x7fffffff;
  if (ix >= 0x4c000000) {
    if (ix > 0x7f800000) return x + x;
    if (hx > 0) return atanhi[3] + atanlo[3];
    return -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3ee00000) {
    if (ix < 0x31000000) return x;
    id = -1;
  } else {
    x = fabsf(x);
    if (ix < 0x3f980000) {
      if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
      else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
    } else {
      if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
      else { id = 3; x = -1.0f / x; }
    }
  }
  z = x * x;
  w = z * z;
  s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
  return (hx < 0) ? -z : z;
}
static float r_atan2f(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  float z;
  int32_t k, m, hx, hy, ix, iy;
  hx = (int32_t)fbits(x); ix = hx & 0x7fffffff;
  hy = (int32_t)fbits(y); iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
  if (hx == 0x3f800000) return r_atanf(y);
  m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
  if (iy == 0) {
    switch (m) { case 0: case 1: return y; case 2: return pi + tiny; default: return -pi - tiny; }
  }
  if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) {
      switch (m) { case 0: return pi_o_4 + tiny; case 1: return -pi_o_4 - tiny; case 2: return 3.0f * pi_o_4 + tiny; default: return -3.0f * pi_o_4 - tiny; }
    } else {
      switch (m) { case 0: return 0.0f; case 1: return -0.0f; case 2: return pi + tiny; default: return -pi - tiny; }
    }
  }
  if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  k = (iy - ix) >> 23;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60) z = 0.0f;
  else z = r_atanf(fabsf(y / x));
  switch (m) {
    case 0: return z;
    case 1: return bitsf(fbits(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}
static inline uint64_t rng(uint64_t* s) { uint64_t x = *s; x ^= x << 13; x ^= x >> 7; x ^= x << 17; return *s = x; }
static inline float u01(uint64_t* s) { return (rng(s) >> 40) * (1.0f / 16777216.0f); }
int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 1000000000L;
  long diff = 0;
#pragma omp parallel reduction(+:diff)
  {
    uint64_t s = 0x9E3779B97F4A7C15ull * (omp_get_thread_num() + 1);
    long per = n / omp_get_num_threads();
    for (long i = 0; i < per; ++i) {
      float dy, dx;
      int fam = i & 3;
      if (fam == 0) {   // box edges
        float cx = (u01(&s) - 0.5f) * 200.f, cy = (u01(&s) - 0.5f) * 200.f;
        float len = 0.2f + u01(&s) * 24.8f, th = (u01(&s) - 0.5f) * 6.2831853f;
        dy = (cy + len * sinf(th)) - cy; dx = (cx + len * cosf(th)) - cx;
      } else if (fam == 1) {   // random bit patterns (all exponents, signs, NaN / Inf / denormals)
        dy = bitsf((uint32_t)rng(&s)); dx = bitsf((uint32_t)rng(&s));
      } else if (fam == 2) {   // near-axis, tiny ratios, exact zeros
        dy = (u01(&s) - 0.5f) * ((rng(&s) & 7) == 0 ? 0.f : ldexpf(1.f, -(int)(rng(&s) % 40)));
        dx = (u01(&s) - 0.5f) * ((rng(&s) & 7) == 0 ? 0.f : 50.f);
      } else {   // ratios next to the argument-reduction thresholds 7/16, 11/16, 19/16, 39/16
        static const float thr[4] = {0.4375f, 0.6875f, 1.1875f, 2.4375f};
        dx = (u01(&s) - 0.5f) * 40.f; dy = dx * thr[rng(&s) & 3] * (1.f + (u01(&s) - 0.5f) * 1e-5f);
        if (rng(&s) & 1) dy = -dy;
      }
      float a = atan2f(dy, dx), b = r_atan2f(dy, dx);
      if (fbits(a) != fbits(b) && !(isnan(a) && isnan(b))) { if (diff < 3) printf("diff y=%a x=%a glibc=%a mine=%a\n", dy, dx, a, b); ++diff; }
    }
  }
  printf("n=%ld bit differences restatement vs glibc atan2f: %ld\n", n, diff);
  return diff != 0;
}
