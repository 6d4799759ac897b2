// ORACLE -- TEST INFRASTRUCTURE ONLY.  Builds the REFERENCE's own weighted NMS
// (/root/reference/operator_cxx/src_cxx/nms.h) as-is into oracle/_ref/, from the sources where they lie.
// No reference source is copied.  nms.h:1 includes overlap.h, which pulls in Eigen through
// pybind11/eigen.h (overlap.h:7); Eigen is absent from this image and nothing wnms_4c uses comes from
// overlap.h, so that one unused include is skipped through its own include guard (overlap.h:1-2) and the
// two declarations it would have contributed (overlap.h:9-10) are stated here.  No stand-in header is
// written for anything.
#define OVERLAP_H
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <iostream>
namespace py = pybind11;
using namespace std;
#include "/root/reference/operator_cxx/src_cxx/nms.h"

static float ref_single_overlap(py::array_t<float> a, py::array_t<float> b, bool is3d) {
  trtplus::OverlapChecker c;
  return c.single_overlap(a.data(), b.data(), is3d);
}
static std::vector<float> ref_pair_overlaps(py::array_t<float> a, py::array_t<float> b, bool is3d) {
  // row-wise IoU of two (n,12) arrays; convenience for bulk pinning
  trtplus::OverlapChecker c;
  size_t n = a.size() / 12;
  std::vector<float> out(n);
  for (size_t i = 0; i < n; ++i) out[i] = c.single_overlap(a.data() + 12 * i, b.data() + 12 * i, is3d);
  return out;
}

PYBIND11_MODULE(processing_cxx_ref, m) {
  m.def("wnms_4c", &point4_wnms_4c<float>);  // same registration as pybinding.cpp:8
  m.def("single_overlap", &ref_single_overlap);
  m.def("pair_overlaps", &ref_pair_overlaps);
}
