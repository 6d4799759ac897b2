// ORACLE -- TEST INFRASTRUCTURE ONLY.  Bulk characterisation of the REFERENCE's BEV overlap routine
// (/root/reference/operator_cxx/src_cxx/nms.h:195-249, compiled as-is from where it lies, exactly like ref_wnms_wrapper.cpp) on
// pairs of boxes that do NOT intersect: the half-plane clipper has no empty-intersection test, so for some disjoint pairs it
// returns NaN, a negative number or -- when two edge directions tie within its EPS = 1e-5 (nms.h:58-64,104-106) -- a positive
// "IoU".  The weighted NMS only compares the result with thresh / thresh_vote (nms.h:509-516), so only positive values matter.
// study_run() draws random disjoint pairs from a named family, runs the reference on each and returns the non-zero results with
// the features a rejection test can be built on.  Built into oracle/_ref/ by `make -C oracle study`; no reference source is copied.
#define OVERLAP_H
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
namespace py = pybind11;
using namespace std;
#include "/root/reference/operator_cxx/src_cxx/nms.h"

namespace {
struct Rng {   // splitmix64 / xorshift: reproducible across platforms
  uint64_t s;
  explicit Rng(uint64_t seed) {   // start points of different seeds are hashed apart (the streams share one 2^64 cycle)
    uint64_t z = (seed + 0x632BE59BD9B4E019ull) * 0xD1342543DE82EF95ull;
    z = (z ^ (z >> 32)) * 0xDABA0B6EB09322E3ull; z = (z ^ (z >> 29)) * 0x9FB21C651E98DF25ull; s = z ^ (z >> 32);
  }
  uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  double u() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
  double uni(double a, double b) { return a + (b - a) * u(); }
  double logu(double a, double b) { return exp(uni(log(a), log(b))); }
};
// the four corners in the order of Decode3DBbox (decode_3d_bbox-inl.h:265-274): A(+l/2,-w/2) B(-l/2,-w/2) C(-l/2,+w/2) D(+l/2,+w/2)
void make_box(double cx, double cy, double l, double w, double yaw, float* d) {
  const double c = cos(yaw), s = sin(yaw);
  const double dx[4] = {l / 2, -l / 2, -l / 2, l / 2}, dy[4] = {-w / 2, -w / 2, w / 2, w / 2};
  for (int k = 0; k < 4; ++k) {
    d[2 * k] = (float)(cx + dx[k] * c - dy[k] * s);
    d[2 * k + 1] = (float)(cy + dx[k] * s + dy[k] * c);
  }
  d[8] = (float)yaw; d[9] = 0.f; d[10] = 1.5f; d[11] = 0.9f;
}
// separating-axis gap of the two quadrilaterals in double precision (> 0: disjoint by that distance along some edge normal)
double sat_gap(const float* a, const float* b) {
  double best = -1e300;
  for (int which = 0; which < 2; ++which) {
    const float* p = which ? b : a;
    for (int e = 0; e < 4; ++e) {
      const double ex = (double)p[2 * ((e + 1) & 3)] - p[2 * e], ey = (double)p[2 * ((e + 1) & 3) + 1] - p[2 * e + 1];
      const double n = sqrt(ex * ex + ey * ey);
      if (n == 0) continue;
      const double nx = ey / n, ny = -ex / n;
      double a0 = 1e300, a1 = -1e300, b0 = 1e300, b1 = -1e300;
      for (int k = 0; k < 4; ++k) {
        const double pa = a[2 * k] * nx + a[2 * k + 1] * ny, pb = b[2 * k] * nx + b[2 * k + 1] * ny;
        a0 = min(a0, pa); a1 = max(a1, pa); b0 = min(b0, pb); b1 = max(b1, pb);
      }
      best = max(best, max(a0 - b1, b0 - a1));
    }
  }
  return best;
}
// smallest |angle_i - angle_j| over the 16 edge pairs, angles as the reference computes them (float atan2 of float differences
// of the corners AFTER its clockwise normalisation -- the reversal negates a direction, i.e. shifts the angle by pi, so both
// orientations of every edge are looked at), with the wrap at +-pi counted as close
double min_edge_angle_diff(const float* a, const float* b) {
  double best = 1e300;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      for (int fa = 0; fa < 2; ++fa)
        for (int fb = 0; fb < 2; ++fb) {
          const int i1 = (i + 1) & 3, j1 = (j + 1) & 3;
          float ax = a[2 * i1] - a[2 * i], ay = a[2 * i1 + 1] - a[2 * i + 1], bx = b[2 * j1] - b[2 * j], by = b[2 * j1 + 1] - b[2 * j + 1];
          if (fa) { ax = -ax; ay = -ay; }
          if (fb) { bx = -bx; by = -by; }
          const float ga = atan2(ay, ax), gb = atan2(by, bx);
          double d = fabs((double)ga - (double)gb);
          d = min(d, fabs(d - 2 * 3.14159265358979323846));
          best = min(best, d);
        }
  return best;
}
// The rejection test of the device pair kernel (rangedet_amd/csrc/k_wnms.h w_box_domain / w_pair_skippable), restated with the same
// float operations on the same 12-float rows: the clip of a pair is skipped only if BOTH boxes are rectangles with edges of
// 0.2 .. 25 m and |coordinates| <= 200 m, their bounding rectangles are more than 0.01 m apart, and their edge directions (mod
// 90 degrees) differ by at least 0.01 rad.
struct BoxDom { bool ok; float x0, y0, x1, y1, phi; };
BoxDom box_domain(const float* c) {
  BoxDom d;
  float ex[4], ey[4], l2[4];
  d.x0 = d.x1 = c[0]; d.y0 = d.y1 = c[1];
  bool ok = true;
  for (int k = 0; k < 4; ++k) {
    const int k1 = (k + 1) & 3;
    ex[k] = c[2 * k1] - c[2 * k]; ey[k] = c[2 * k1 + 1] - c[2 * k + 1];
    l2[k] = ex[k] * ex[k] + ey[k] * ey[k];
    ok = ok && l2[k] >= 0.04f && l2[k] <= 625.f && fabsf(c[2 * k]) <= 200.f && fabsf(c[2 * k + 1]) <= 200.f;
    d.x0 = fminf(d.x0, c[2 * k]); d.x1 = fmaxf(d.x1, c[2 * k]); d.y0 = fminf(d.y0, c[2 * k + 1]); d.y1 = fmaxf(d.y1, c[2 * k + 1]);
  }
  // a rectangle up to rounding: opposite edges cancel, adjacent edges are perpendicular (1e-3 relative)
  const float t0 = 1e-3f * sqrtf(l2[0]), t1 = 1e-3f * sqrtf(l2[1]);
  ok = ok && fabsf(ex[0] + ex[2]) <= t0 && fabsf(ey[0] + ey[2]) <= t0 && fabsf(ex[1] + ex[3]) <= t1 && fabsf(ey[1] + ey[3]) <= t1;
  ok = ok && fabsf(ex[0] * ex[1] + ey[0] * ey[1]) <= 1e-3f * sqrtf(l2[0] * l2[1]);
  float a = atan2f(ey[0], ex[0]);                        // direction of one edge, folded to [0, pi/2)
  a = a < 0.f ? a + 3.14159265f : a;
  a = a >= 1.57079633f ? a - 1.57079633f : a;
  d.phi = a; d.ok = ok && a == a;
  return d;
}
bool pair_skippable(const BoxDom& a, const BoxDom& b) {
  if (!a.ok || !b.ok) return false;
  const bool apart = a.x1 + 0.01f < b.x0 || b.x1 + 0.01f < a.x0 || a.y1 + 0.01f < b.y0 || b.y1 + 0.01f < a.y0;
  float d = fabsf(a.phi - b.phi);
  d = fminf(d, 1.57079633f - d);
  return apart && d >= 0.01f;
}
}  // namespace

// family: 0 general position (uniform yaw), 1 relative yaw = k*pi/2 + eps with eps log-uniform in [eps_lo, eps_hi] (random sign),
// 2 exactly equal yaw parameters, 3 nearly touching (SAT gap log-uniform in [1e-6, 1e-2]), 4 extreme sizes / far from the origin
// Returns rows [a(12) | b(12) | ovr | sat_gap | min_edge_angle_diff] for every pair whose reference result is not exactly 0, plus the
// count of pairs evaluated; pairs are kept only if their SAT gap is > min_gap (i.e. they are disjoint).
static py::tuple study_run(int family, uint64_t seed, long n, double eps_lo, double eps_hi, double min_gap, bool keep_all) {
  std::vector<float> rows;
  long evaluated = 0, nnan = 0, nneg = 0, npos = 0, nharm = 0, nskip = 0, nviol = 0;
  {
  py::gil_scoped_release nogil;   // (the loop touches no Python object: several threads of one process can run it)
  Rng r(seed);
  trtplus::OverlapChecker c;
  float a[12], b[12];
  for (long it = 0; it < n; ++it) {
    double cx = r.uni(-75, 75), cy = r.uni(-75, 75);
    double l1 = r.uni(0.5, 12), w1 = r.uni(0.3, 3.5), l2 = r.uni(0.5, 12), w2 = r.uni(0.3, 3.5);
    double y1 = r.uni(-M_PI, M_PI), y2 = r.uni(-M_PI, M_PI);
    double dist = r.uni(0, 30), th = r.uni(-M_PI, M_PI);
    if (family == 1) {
      const double e = r.logu(eps_lo, eps_hi) * (r.u() < 0.5 ? -1 : 1);
      y2 = y1 + floor(r.uni(0, 4)) * (M_PI / 2) + e;
    } else if (family == 2) {
      y2 = y1 + floor(r.uni(0, 4)) * (M_PI / 2);
      if (r.u() < 0.5) { y1 = floor(r.uni(-2, 3)) * (M_PI / 2); y2 = y1 + floor(r.uni(0, 4)) * (M_PI / 2); }   // axis aligned
    } else if (family == 4) {
      const int k = (int)r.uni(0, 4);
      if (k == 0) { cx *= 20; cy *= 20; }                          // +-1500 m
      if (k == 1) { l1 = r.logu(0.01, 0.5); w1 = r.logu(0.01, 0.5); dist = r.uni(0, 3); }
      if (k == 2) { l2 = r.logu(12, 200); w2 = r.logu(0.01, 0.3); dist = r.uni(0, 120); }
      if (k == 3) { l1 = r.logu(0.01, 200); w1 = r.logu(0.01, 200); l2 = r.logu(0.01, 200); w2 = r.logu(0.01, 200); dist = r.uni(0, 300); }
    }
    if (family >= 5) {
      // the DOMAIN of the rejection test (rd_wnms_4c skips the clip of a pair only inside it): edges 0.2 .. 40 m, |coordinate| <= 200 m,
      // sampled log-uniformly in size, close together, with every relative orientation (5) or nearly parallel (6: eps_lo .. eps_hi)
      cx = r.uni(-185, 185); cy = r.uni(-185, 185);
      l1 = r.logu(0.2, 25); w1 = r.logu(0.2, 25); l2 = r.logu(0.2, 25); w2 = r.logu(0.2, 25);
      if (family == 8) {   // traffic: car / truck sized boxes aligned with a road direction up to a few degrees, +-80 m
        cx = r.uni(-80, 80); cy = r.uni(-80, 80);
        l1 = r.uni(3, 12); w1 = r.uni(1.5, 3); l2 = r.uni(3, 12); w2 = r.uni(1.5, 3);
        const double road = r.uni(-M_PI, M_PI);
        y1 = road + floor(r.uni(0, 4)) * (M_PI / 2) + r.uni(-0.08, 0.08); y2 = road + floor(r.uni(0, 4)) * (M_PI / 2) + r.uni(-0.08, 0.08);
      }
      const double reach = 0.5 * (sqrt(l1 * l1 + w1 * w1) + sqrt(l2 * l2 + w2 * w2));
      dist = r.uni(0, 1.3 * reach);
      if (family == 6) {
        const double e = r.logu(eps_lo, eps_hi) * (r.u() < 0.5 ? -1 : 1);
        y2 = y1 + floor(r.uni(0, 4)) * (M_PI / 2) + e;
      }
      if (family == 7) {   // axis-parallel to the coordinate axes up to eps (edge angles next to 0, +-pi/2, +-pi: the atan2 wrap)
        y1 = floor(r.uni(-2, 3)) * (M_PI / 2) + r.logu(eps_lo, eps_hi) * (r.u() < 0.5 ? -1 : 1);
      }
    }
    make_box(cx, cy, l1, w1, y1, a);
    if (family == 3) {   // slide the second box along th until the SAT gap is tiny
      double lo = 0, hi = 400;
      const double want = r.logu(1e-6, 1e-2);
      for (int k = 0; k < 60; ++k) {
        const double mid = 0.5 * (lo + hi);
        make_box(cx + mid * cos(th), cy + mid * sin(th), l2, w2, y2, b);
        if (sat_gap(a, b) > want) hi = mid; else lo = mid;
      }
      dist = hi;
    }
    make_box(cx + dist * cos(th), cy + dist * sin(th), l2, w2, y2, b);
    const double gap = sat_gap(a, b);
    if (!(gap > min_gap)) continue;
    ++evaluated;
    const float o = c.single_overlap(a, b, false);
    const bool skip = pair_skippable(box_domain(a), box_domain(b));
    if (skip) ++nskip;
    if (o == 0.f) continue;
    if (o != o) ++nnan; else if (o < 0) ++nneg; else { ++npos; if (o >= 0.05f) ++nharm; }
    if (skip && o >= 1e-6f) ++nviol;       // a skipped pair whose reference result could pass a threshold >= 1e-3 (guard: 1e-6)
    if (o == o && o > 0 && (keep_all || skip)) {   // only positive results are kept row by row (NaN / negative never pass a >= / > comparison)
      rows.insert(rows.end(), a, a + 12);
      rows.insert(rows.end(), b, b + 12);
      rows.push_back(o); rows.push_back((float)gap); rows.push_back((float)min_edge_angle_diff(a, b));
    }
  }
  }
  py::array_t<float> out(rows.size());
  memcpy(out.mutable_data(), rows.data(), rows.size() * sizeof(float));
  return py::make_tuple(out, evaluated, nnan, nneg, npos, nharm, nskip, nviol);
}

PYBIND11_MODULE(ref_overlap_study, m) { m.def("study_run", &study_run); }
