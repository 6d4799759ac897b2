// Development harness of the Winograd F(2,3) 3x3 conv experiment (tools/micro/k_wino.h) as its own small translation unit: builds in
// seconds for the GPU (hipcc --offload-arch=gfx950 -O3 -shared -fPIC) and for the CPU emulator (tests/emu shim); tools/micro/wino_dev.py
// drives both.
#include "k_wino.h"
extern "C" {
size_t rdm_wino_packed_bytes(int cin) { return rd::wino_packed_bytes(cin); }
// w: (128, cin, 3, 3) fp32, scale (128) or null folded into the weights
int rdm_pack_wino_host(const float* w, const float* scale, int cin, int dtype, void* out) {
  rd::pack_wino_frag(cin, out, [&](int co, int ci, int dh, int dw) { return (scale ? scale[co] : 1.f) * w[(((size_t)co * cin + ci) * 3 + dh) * 3 + dw]; }, dtype);
  return 0;
}
int rdm_wino(const void* x, int x_cs, int x_co, const void* w, const float* shift, const void* res, int r_cs, int r_co, void* y, int y_cs,
             int y_co, int B, int H, int W, int cin, int flags, int dtype, void* stream) {
  return rd::launch_wino(x, x_cs, x_co, w, shift, res, r_cs, r_co, y, y_cs, y_co, B, H, W, cin, flags, dtype, (hipStream_t)stream);
}
const char* rdm_last_error(void) { return rd::err_buf(); }
}
