// Development harness of the fused BasicBlock kernel in its 16 x 16 x 32 MFMA form (k_block.h M16, round 6) as its own small translation unit
// (seconds to build): register allocation and timing against the 32 x 32 x 16 form; tools/micro/block16_bench.py drives it.
#define RD_CONV3_DEV_M16_ONLY
#include "../../rangedet_amd/csrc/k_block.h"
extern "C" {
size_t rdm_block_packed_bytes(void) { return rd::block64_packed_bytes(64); }
int rdm_pack_block(const float* w1, const float* s1, const float* w2, const float* s2, int m16, void* out) {
  if (m16) rd::pack_block64_m16(w1, s1, w2, s2, RD_BF16, out); else rd::pack_block64(w1, s1, w2, s2, 64, RD_BF16, out);
  return 0;
}
int rdm_pack_sc(const float* w, const float* s, int m16, void* out) {
  if (m16) rd::pack_sc_frag16(w, s, out, RD_BF16); else rd::pack_sc_frag(w, s, 64, 64, out, RD_BF16);
  return 0;
}
int rdm_block(const void* x, const void* w, const float* shift1, const float* shift2, const void* sc_w, void* y, int B, int H, int W, int m16, void* stream) {
  return rd::launch_block64(x, 64, 0, 64, w, shift1, shift2, sc_w, y, 64, 0, B, H, W, RD_BF16, (hipStream_t)stream, m16 != 0);
}
const char* rdm_last_error(void) { return rd::err_buf(); }
}
