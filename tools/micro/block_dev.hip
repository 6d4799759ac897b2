// Development harness of the fused BasicBlock kernel (k_block.h) as its own small translation unit: builds in seconds for the GPU
// (hipcc --offload-arch=gfx950 -O3 -shared -fPIC) and for the CPU emulator (tests/emu shim), tools/micro/block_dev.py drives both.
#include "../../rangedet_amd/csrc/k_block.h"
#include "k_block16.h"
extern "C" {
size_t rdm_block64_packed_bytes(int cin) { return rd::block64_packed_bytes(cin); }
int rdm_pack_block64_host(const float* w1, const float* s1, const float* w2, const float* s2, int cin, int dtype, void* out) {
  rd::pack_block64(w1, s1, w2, s2, cin, dtype, out);
  return 0;
}
int rdm_block64(const void* x, int x_cs, int x_co, int cin, const void* w, const float* shift1, const float* shift2, const void* sc_w, void* y,
                int y_cs, int y_co, int B, int H, int W, int dtype, void* stream) {
  return rd::launch_block64(x, x_cs, x_co, cin, w, shift1, shift2, sc_w, y, y_cs, y_co, B, H, W, dtype, (hipStream_t)stream);
}
int rdm_block64_tall(const void* x, int x_cs, int x_co, const void* w, const float* shift1, const float* shift2, void* y, int y_cs, int y_co,
                     int B, int H, int W, int dtype, void* stream) {
  return rd::launch_block64_tall(x, x_cs, x_co, w, shift1, shift2, y, y_cs, y_co, B, H, W, dtype, (hipStream_t)stream);
}
const char* rdm_last_error(void) { return rd::err_buf(); }
}
