// Test-time input transform chain of the reference on the device: from the raw record
// arrays straight to the named tensors the graph consumes, one fused gather kernel instead of ~10 numpy passes.
//   LoadRecord            rangedet/core/input.py:14-42     mask = range > 0, pc zeroed where invalid
//   ProcessMissValue      input.py:89-137                  1-px azimuth fill of -1 returns, then "still missing" -> far
//                                                          return (80,0,0,-1) or, next to a valid return (+-2 px), a
//                                                          car-window return (0,0,0,-1)
//   SepAndClipData / GetUnnormalizedRange / NormData / GetCoordinates / CombineData     input.py:140-229
//   PadData / TransposeData               input.py:522-558
//   GenerateFPNTarget / TransAndReshape   input.py:561-624  per-level range-interval masks, strided sampling s//2::s
// One thread per padded pixel; every value it needs is a pure function of <= 10 raw returns (its own, the one to its
// right, and the 4 returns two pixels away with THEIR right neighbours), all wrap-around like the reference's index
// lists.  HBM-bound: 28 B/px in, ~100 B/px out.
#pragma once
#include "rd_common.h"
#include "../../include/rangedet_hip.h"

namespace rd {

struct InputArgs {
  const float* ri;    // (B,H,W,4) range, intensity, elongation, (unused)
  const float* pc;    // (B,H,W,3) vehicle-frame xyz
  const float* incl;  // (B,H)
  float* data;        // (B,8,Hp,Wp)
  float* coord;       // (B,3,Hp,Wp)
  float* pcs[3];      // (B,Hp*Wp/s,3) for s = 1,2,4
  float* msk[3];      // (B,Hp*Wp/s)
  int B, H, W, Hp, Wp;
  rd_input_norm_t n;
};

__device__ __forceinline__ float in_norm(float v, float lo, float hi, float mean, float sd, bool clip) {
  if (clip) v = fminf(fmaxf(v, lo), hi);
  return (v - mean) / sd;
}

__global__ __launch_bounds__(256) void input_transform_kernel(InputArgs a) {
  const long npx = (long)a.Hp * a.Wp;
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= npx) return;
  const int b = blockIdx.y;
  const int h = (int)(i / a.Wp), w = (int)(i - (long)h * a.Wp);
  const float* ri = a.ri + (size_t)b * a.H * a.W * 4;
  const float* pc = a.pc + (size_t)b * a.H * a.W * 3;
  float f[8];                       // range, intensity, elongation, x, y, z (raw after the miss-value pass)
  float rmask = 0.f, unnorm = 0.f;
  float d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float px = 0.f, py = 0.f, pz = 0.f;
  if (h < a.H && w < a.W) {
    // range value after the 1-px fill at any (hh, ww)
    auto r0f = [&](int hh, int ww) {
      const float r = ri[((size_t)hh * a.W + ww) * 4];
      return r == -1.f ? ri[((size_t)hh * a.W + (ww + 1 == a.W ? 0 : ww + 1)) * 4] : r;
    };
    const float r_here = ri[((size_t)h * a.W + w) * 4];
    const int ws = r_here == -1.f ? (w + 1 == a.W ? 0 : w + 1) : w;      // source column of this pixel's values
    const float* rs = ri + ((size_t)h * a.W + ws) * 4;
    const float* ps = pc + ((size_t)h * a.W + ws) * 3;
    const bool valid_src = rs[0] > 0.f;                                   // LoadRecord mask of the source return
    f[0] = rs[0]; f[1] = rs[1]; f[2] = rs[2];
    px = valid_src ? ps[0] : 0.f; py = valid_src ? ps[1] : 0.f; pz = valid_src ? ps[2] : 0.f;
    rmask = valid_src ? 1.f : 0.f;
    if (f[0] == -1.f) {                                                   // still missing after the fill
      const int hd = h >= 2 ? h - 2 : h - 2 + a.H, hu = h + 2 < a.H ? h + 2 : h + 2 - a.H;
      const int wr = w >= 2 ? w - 2 : w - 2 + a.W, wl = w + 2 < a.W ? w + 2 : w + 2 - a.W;
      const bool car = r0f(hd, w) != -1.f || r0f(hu, w) != -1.f || r0f(h, wr) != -1.f || r0f(h, wl) != -1.f;
      f[0] = car ? 0.f : 80.f; f[1] = 0.f; f[2] = 0.f;
      px = py = pz = 0.f;
    }
    const float az = atan2f(py, px);
    const float inc = a.incl[(size_t)b * a.H + h];
    const rd_input_norm_t& n = a.n;
    d[0] = in_norm(f[0], n.clip_lo[0], n.clip_hi[0], n.mean[0], n.sd[0], true);
    unnorm = fminf(fmaxf(f[0], n.clip_lo[0]), n.clip_hi[0]);
    d[1] = in_norm(f[1], n.clip_lo[1], n.clip_hi[1], n.mean[1], n.sd[1], true);
    d[2] = in_norm(f[2], n.clip_lo[2], n.clip_hi[2], n.mean[2], n.sd[2], true);
    d[3] = in_norm(px, n.clip_lo[3], n.clip_hi[3], n.mean[3], n.sd[3], true);
    d[4] = in_norm(py, n.clip_lo[4], n.clip_hi[4], n.mean[4], n.sd[4], true);
    d[5] = in_norm(pz, n.clip_lo[5], n.clip_hi[5], n.mean[5], n.sd[5], true);
    d[6] = in_norm(inc, n.clip_lo[6], n.clip_hi[6], n.mean[6], n.sd[6], true);
    d[7] = in_norm(az, 0.f, 0.f, n.mean[7], n.sd[7], false);              // azimuth is not clipped (input.py:149)
  }
  float* dp = a.data + (size_t)b * 8 * npx + i;
#pragma unroll
  for (int c = 0; c < 8; ++c) dp[(size_t)c * npx] = d[c];
  float* cp = a.coord + (size_t)b * 3 * npx + i;
  cp[0] = d[3]; cp[npx] = d[4]; cp[2 * npx] = d[5];
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const int s = 1 << l;
    if ((w & (s - 1)) != (s >> 1)) continue;                              // sampled columns s//2, s//2 + s, ...
    const long j = (long)h * (a.Wp / s) + (w >> l);
    const long ns = npx / s;
    const float m = (a.n.interval_lo[l] <= unnorm && unnorm < a.n.interval_hi[l]) ? 1.f : 0.f;
    a.msk[l][(size_t)b * ns + j] = rmask * m;
    float* q = a.pcs[l] + ((size_t)b * ns + j) * 3;
    q[0] = px; q[1] = py; q[2] = pz;
  }
}

}  // namespace rd
