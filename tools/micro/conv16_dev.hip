// Development harness of the persistent 3x3 conv (k_conv3.h) as its own small translation unit, for the v_mfma_f32_16x16x32 form (M16, round 6):
// builds in under a minute for the GPU (hipcc --offload-arch=gfx950 -O3 -shared -fPIC -Xarch_device -fno-slp-vectorize), so that register
// allocation and timing of ONE form can be iterated without the 4-minute library build; tools/micro/conv16_bench.py drives it.
#define RD_CONV3_DEV_M16_ONLY
#include "../../rangedet_amd/csrc/k_conv3.h"
extern "C" {
size_t rdm_conv3_packed_bytes(int cin, int cout) { return rd::conv_packed_bytes(9, cin, cout, RD_BF16); }
int rdm_pack_conv3(const float* w, const float* fs, int cout, int cin, int m16, int dtype, void* out) {
  memset(out, 0, rdm_conv3_packed_bytes(cin, cout));
  auto get = [&](int co, int ci, int t) { return (fs ? fs[co] : 1.f) * w[(((size_t)co * cin + ci) * 3 + t / 3) * 3 + t % 3]; };
  if (m16) rd::pack_taps_frag16(9, cin, cout, out, get, dtype);
  else rd::pack_taps_frag(9, cin, cout, out, get, dtype);
  return 0;
}
int rdm_conv3(const void* x, int x_cs, const void* w, const float* shift, const void* res, void* y, int B, int H, int W, int cin, int cout,
              int flags, int m16, int dtype, void* stream) {
  return rd::launch_conv3(x, x_cs, 0, w, nullptr, shift, res, cout, 0, y, cout, 0, B, H, W, cin, cout, flags | RD_SCALE_FOLDED, 1, (hipStream_t)stream, 0,
                          nullptr, dtype, nullptr, nullptr, nullptr, m16 ? rd::C3_BODY_M16 : 0);
}
// the M16 form with the fused 1x1 output conv (head weights: pack_head_frag16)
int rdm_pack_head16(const float* w, int nout, int cin, int dtype, void* out) { rd::pack_head_frag16(w, nout, cin, out, dtype); return 0; }
int rdm_conv3_head(const void* x, int x_cs, const void* w, const float* shift, const void* hw, const float* hb, float* out, int nout, int B, int H,
                   int W, int cin, int dtype, void* stream) {
  rd::Conv3Args h;
  memset(&h, 0, sizeof(h));
  h.hw = (const unsigned char*)hw; h.hb = hb; h.ho = out; h.ho_bs = (long)H * W * nout; h.ho_off = 0; h.hn = nout;
  return rd::launch_conv3(x, x_cs, 0, w, nullptr, shift, nullptr, 0, 0, nullptr, 128, 0, B, H, W, cin, 128, RD_RELU_POST | RD_SCALE_FOLDED, 1,
                          (hipStream_t)stream, 0, &h, dtype, nullptr, nullptr, nullptr, rd::C3_BODY_M16);
}
const char* rdm_last_error(void) { return rd::err_buf(); }
}
