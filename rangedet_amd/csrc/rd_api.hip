// librangedet_hip.so -- C ABI (include/rangedet_hip.h) over the hand-written gfx950 kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared rd_api.hip -o librangedet_hip.so
#include "k_conv1.h"
#include "k_block.h"
#include "k_assign.h"
#include "k_input.h"
#include "k_nms3d.h"
#include "k_meta.h"
#include "k_misc.h"
#include "k_riou.h"
#include "k_sort.h"
#include "k_wnms.h"

#include <algorithm>
#include <string>
#include <numeric>

using namespace rd;

namespace rd {
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
inline size_t wnms_ws_bytes(int cap) {
  const size_t nw = (size_t)(cap + 63) / 64;
  return align256((size_t)cap * PREP_F * 4) + 3 * align256((size_t)cap * nw * 8) + 4 * align256((size_t)cap * 4) +
         align256(((size_t)cap + 2) * 8 + 3 * ((size_t)cap / 32 + 2) * 4) + align256(nw * 8) + 256 + align256(sort_ws_bytes(cap)) + 256;
}
inline WnmsWs wnms_ws_carve(void* ws, int cap) {
  WnmsWs w;
  const size_t nw = (size_t)(cap + 63) / 64;
  unsigned char* p = (unsigned char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  w.nwcap = (int)nw;
  w.prep = (float*)p; p += align256((size_t)cap * PREP_F * 4);
  w.thr = (unsigned long long*)p; p += align256((size_t)cap * nw * 8);
  w.vote = (unsigned long long*)p; p += align256((size_t)cap * nw * 8);
  w.snap = (unsigned long long*)p; p += align256((size_t)cap * nw * 8);
  w.keep_q = (int*)p; p += align256((size_t)cap * 4);
  w.order = (int*)p; p += align256((size_t)cap * 4);
  w.alive = (int*)p; p += align256((size_t)cap * 4);
  w.ovf = (int*)p; p += align256((size_t)cap * 4);
  // merge overflow list / tie-order keys (cap + 2 ints + cap + 2 floats, or cap floats + cap ints + a cap-bit map)
  w.scratch = (int*)p; p += align256(((size_t)cap + 2) * 8 + 3 * ((size_t)cap / 32 + 2) * 4);
  w.supp_state = (unsigned long long*)p; p += align256(nw * 8);
  w.nalive = (int*)p; w.novf = (int*)p + 1; p += 256;
  w.sort_ws = p;
  return w;
}
inline void allow_conv_lds() {   // the generic tap kernel (the persistent 3x3 kernel does this per instantiation, k_conv3.h c3_go)
  static std::atomic<unsigned long long> seen{0};
  once_per_device(seen, [] {
    allow_big_lds(conv_taps_kernel<RD_BF16, 4, 3>); allow_big_lds(conv_taps_kernel<RD_BF16, 4, 8>);
    allow_big_lds(conv_taps_kernel<RD_F16, 4, 3>); allow_big_lds(conv_taps_kernel<RD_F16, 4, 8>);
    allow_big_lds(conv_taps_kernel<RD_F32, 4, 8>); allow_big_lds(conv_taps_kernel<RD_F32, 4, 3>);
  });
}
}  // namespace rd

extern "C" {

int rd_version(void) { return 100; }
const char* rd_last_error_string(void) { return rd::err_buf(); }

// ---- layout ---------------------------------------------------------------------------------------------
int rd_nchw_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int dst_cstride, int dst_coff,
                    int zero_pad, int dst_dtype, void* stream) {
  RD_REQUIRE(src && dst, RD_EINVAL, "nchw_to_nhwc: null pointer");
  RD_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && zero_pad >= 0, RD_ESHAPE, "nchw_to_nhwc: bad shape");
  RD_REQUIRE(dst_coff >= 0 && dst_coff + C + zero_pad <= dst_cstride, RD_ESHAPE, "nchw_to_nhwc: channels exceed stride");
  hipStream_t st = (hipStream_t)stream;
  const long HW = (long)H * W, total = (long)B * (C + zero_pad) * HW;
  const int grid = (int)std::min<long>((total + 255) / 256, 8192);
  ProfScope ps(RD_PROF_LAYOUT, st);
  {  // few channels, slot-aligned destination: pixel-per-thread variant with 16-byte stores
    const int ch = ch_per_slot(dst_dtype), ctot = C + zero_pad;
    if ((is_h16(dst_dtype) || dst_dtype == RD_F32) && ctot % ch == 0 && ctot / ch <= 2 && dst_cstride % ch == 0 && dst_coff % ch == 0) {
      const long npix = (long)B * HW;
      const int g2 = (int)std::min<long>((npix + 255) / 256, 8192);
#define RD_PX(DT, NS) hipLaunchKernelGGL((nchw_to_nhwc_px_kernel<DT, NS>), dim3(g2), dim3(256), 0, st, src, dst, C, HW, dst_cstride, dst_coff, npix)
      if (dst_dtype == RD_BF16) { if (ctot / ch == 1) RD_PX(RD_BF16, 1); else RD_PX(RD_BF16, 2); }
      else if (dst_dtype == RD_F16) { if (ctot / ch == 1) RD_PX(RD_F16, 1); else RD_PX(RD_F16, 2); }
      else { if (ctot / ch == 1) RD_PX(RD_F32, 1); else RD_PX(RD_F32, 2); }
#undef RD_PX
      return check_launch("nchw_to_nhwc");
    }
  }
  if (dst_dtype == RD_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<RD_BF16>, dim3(grid), dim3(256), 0, st, src, dst, C, HW, dst_cstride, dst_coff, zero_pad, total);
  else if (dst_dtype == RD_F16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<RD_F16>, dim3(grid), dim3(256), 0, st, src, dst, C, HW, dst_cstride, dst_coff, zero_pad, total);
  else if (dst_dtype == RD_F32)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<RD_F32>, dim3(grid), dim3(256), 0, st, src, dst, C, HW, dst_cstride, dst_coff, zero_pad, total);
  else
    return rd::fail(RD_EINVAL, "nchw_to_nhwc: dtype %d", dst_dtype);
  return check_launch("nchw_to_nhwc");
}
int rd_copy_rows(const void* src, long src_row_bytes, void* dst, long dst_row_bytes, long dst_offset_bytes, long bytes,
                 int rows, void* stream) {
  RD_REQUIRE(src && dst, RD_EINVAL, "copy_rows: null pointer");
  RD_REQUIRE(rows > 0 && bytes > 0 && src_row_bytes >= bytes && dst_row_bytes >= dst_offset_bytes + bytes && dst_offset_bytes >= 0,
             RD_ESHAPE, "copy_rows: bad geometry");
  ProfScope ps(RD_PROF_LAYOUT, (hipStream_t)stream);
  if (hipMemcpy2DAsync((char*)dst + dst_offset_bytes, (size_t)dst_row_bytes, src, (size_t)src_row_bytes, (size_t)bytes,
                       (size_t)rows, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
    return rd::fail(RD_EHIP, "copy_rows: hipMemcpy2DAsync failed");
  return RD_OK;
}
int rd_nhwc_to_nchw(const void* src, float* dst, int B, int C, int H, int W, int src_cstride, int src_coff,
                    int src_dtype, void* stream) {
  RD_REQUIRE(src && dst, RD_EINVAL, "nhwc_to_nchw: null pointer");
  RD_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, RD_ESHAPE, "nhwc_to_nchw: bad shape");
  RD_REQUIRE(src_coff >= 0 && src_coff + C <= src_cstride, RD_ESHAPE, "nhwc_to_nchw: channels exceed stride");
  hipStream_t st = (hipStream_t)stream;
  const long HW = (long)H * W, total = (long)B * C * HW;
  const int grid = (int)std::min<long>((total + 255) / 256, 8192);
  ProfScope ps(RD_PROF_LAYOUT, st);
  if (src_dtype == RD_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<RD_BF16>, dim3(grid), dim3(256), 0, st, src, dst, C, HW, src_cstride, src_coff, total);
  else if (src_dtype == RD_F16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<RD_F16>, dim3(grid), dim3(256), 0, st, src, dst, C, HW, src_cstride, src_coff, total);
  else if (src_dtype == RD_F32)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<RD_F32>, dim3(grid), dim3(256), 0, st, src, dst, C, HW, src_cstride, src_coff, total);
  else
    return rd::fail(RD_EINVAL, "nhwc_to_nchw: dtype %d", src_dtype);
  return check_launch("nhwc_to_nchw");
}

// ---- packing (host) -------------------------------------------------------------------------------------
size_t rd_conv_packed_bytes(int ntaps, int cin, int cout, int dtype) { return conv_packed_bytes(ntaps, cin, cout, dtype); }
int rd_pack_conv_weight_host(const float* w, int cout, int cin, int kh, int kw, int dtype, void* out) {
  RD_REQUIRE(w && out, RD_EINVAL, "pack_conv: null pointer");
  RD_REQUIRE(dtype == RD_F32 || is_h16(dtype), RD_EINVAL, "pack_conv: dtype");
  RD_REQUIRE(kh * kw >= 1 && kh * kw <= 9 && (kh & 1) && (kw & 1), RD_ESHAPE, "pack_conv: kernel (%d,%d)", kh, kw);
  TapList tl = conv_taps(kh, kw);
  auto get = [&](int co, int ci, int t) { return w[(((size_t)co * cin + ci) * kh + tl.kh[t]) * kw + tl.kw[t]]; };
  memset(out, 0, conv_packed_bytes(tl.n, cin, cout, dtype));     // padding of partial chunks and the zero tail
  if (is_h16(dtype)) pack_taps_frag(tl.n, cin, cout, out, get, dtype);
  else pack_taps(tl.n, cin, cout, dtype, out, get);
  return RD_OK;
}
// A phase whose taps all lie inside the 3x3 window (every transposed conv of the RangeDet graph) is packed and run as a
// tap list in (dh, dw) order; the persistent 3x3 kernel then serves it, writing the phase's pixels of the output viewed
// as [H][Win][stride_w * Cstride].  Tap set 1 / 2: exactly the six taps with dh in {-1,0,1} and dw in {-1,0} / {0,+1}
// (what k(3,8) s4 p2 and k(3,4) s2 p1 phases are) -> 6-step units; anything else inside the window is embedded in
// all nine taps with zero weights.
static bool deconv_embeds_3x3(const TapList& tl) {
  for (int t = 0; t < tl.n; ++t)
    if (tl.dh[t] < -1 || tl.dh[t] > 1 || tl.dw[t] < -1 || tl.dw[t] > 1) return false;
  return tl.n >= 1 && tl.n <= 9;
}
static int deconv_tap_set(const TapList& tl) {   // 1 / 2 as above, else 0
  if (tl.n != 6 || !deconv_embeds_3x3(tl)) return 0;
  for (int ts = 1; ts <= 2; ++ts) {
    bool seen[6] = {false, false, false, false, false, false};
    bool ok = true;
    for (int t = 0; t < 6 && ok; ++t) {
      const int c = tl.dw[t] - (ts == 1 ? -1 : 0);
      if (c < 0 || c > 1) ok = false;
      else seen[(tl.dh[t] + 1) * 2 + c] = true;
    }
    for (int i = 0; i < 6 && ok; ++i) ok = seen[i];
    if (ok) return ts;
  }
  return 0;
}
// the phase's taps in ascending (dh, dw) order: the order the packed image and every kernel use
static TapList deconv_taps_sorted(int kh, int kw, int s, int pad_w, int phase) {
  TapList tl = deconv_taps(kh, kw, s, pad_w, phase);
  if (tl.n > 9) return tl;
  for (int i = 1; i < tl.n; ++i)
    for (int j = i; j > 0 && (tl.dh[j] < tl.dh[j - 1] || (tl.dh[j] == tl.dh[j - 1] && tl.dw[j] < tl.dw[j - 1])); --j) {
      std::swap(tl.dh[j], tl.dh[j - 1]); std::swap(tl.dw[j], tl.dw[j - 1]);
      std::swap(tl.kh[j], tl.kh[j - 1]); std::swap(tl.kw[j], tl.kw[j - 1]);
    }
  return tl;
}
int rd_deconv_phase_taps(int kh, int kw, int stride_w, int pad_w, int phase) {
  if (stride_w < 1 || phase < 0 || phase >= stride_w) return RD_EINVAL;
  TapList tl = deconv_taps(kh, kw, stride_w, pad_w, phase);
  if (tl.n > 9) return RD_ESHAPE;
  return (deconv_embeds_3x3(tl) && !deconv_tap_set(tl)) ? 9 : tl.n;
}
int rd_pack_deconv_weight_host(const float* w, int cin, int cout, int kh, int kw, int stride_w, int pad_w,
                               int phase, int dtype, void* out) {
  return rd_pack_deconv_weight_folded_host(w, nullptr, cin, cout, kh, kw, stride_w, pad_w, phase, dtype, out);
}
int rd_pack_deconv_weight_folded_host(const float* w, const float* fold_scale, int cin, int cout, int kh, int kw, int stride_w,
                                      int pad_w, int phase, int dtype, void* out) {
  RD_REQUIRE(w && out, RD_EINVAL, "pack_deconv: null pointer");
  RD_REQUIRE(dtype == RD_F32 || is_h16(dtype), RD_EINVAL, "pack_deconv: dtype");
  RD_REQUIRE(stride_w >= 1 && phase >= 0 && phase < stride_w, RD_EINVAL, "pack_deconv: phase");
  TapList tl = deconv_taps_sorted(kh, kw, stride_w, pad_w, phase);
  RD_REQUIRE(tl.n >= 1 && tl.n <= 9, RD_ESHAPE, "pack_deconv: %d taps", tl.n);
  const bool emb = deconv_embeds_3x3(tl) && !deconv_tap_set(tl);
  auto get = [&](int co, int ci, int t) -> float {
    if (emb) {  // t indexes the 3x3 window (dh, dw) = (t/3 - 1, t%3 - 1): the phase's tap there, or zero
      int src = -1;
      for (int u = 0; u < tl.n; ++u)
        if (tl.dh[u] == t / 3 - 1 && tl.dw[u] == t % 3 - 1) src = u;
      if (src < 0) return 0.f;
      t = src;
    }
    return (fold_scale ? fold_scale[co] : 1.f) * w[(((size_t)ci * cout + co) * kh + tl.kh[t]) * kw + tl.kw[t]];
  };
  const int nt = emb ? 9 : tl.n;
  memset(out, 0, conv_packed_bytes(nt, cin, cout, dtype));
  if (is_h16(dtype)) pack_taps_frag(nt, cin, cout, out, get, dtype);
  else pack_taps(nt, cin, cout, dtype, out, get);
  return RD_OK;
}

// ---- conv family ------------------------------------------------------------------------------------------
int rd_conv2d_bn_act(const void* x, int x_cstride, int x_coff, const void* w_packed, const float* scale,
                     const float* shift, const void* residual, int r_cstride, int r_coff, void* y, int y_cstride,
                     int y_coff, int B, int H, int Win, int cin, int cout, int kh, int kw, int stride_w, int flags,
                     int dtype, void* stream) {
  RD_REQUIRE(x && w_packed && y, RD_EINVAL, "conv2d: null pointer");
  RD_REQUIRE((kh == 1 && kw == 1) || (kh == 3 && kw == 3), RD_ESHAPE, "conv2d: kernel (%d,%d) not in {1x1,3x3}", kh, kw);
  RD_REQUIRE(stride_w == 1 || stride_w == 2, RD_ESHAPE, "conv2d: stride_w %d", stride_w);
  RD_REQUIRE(y_coff >= 0 && y_coff + cout <= y_cstride, RD_ESHAPE, "conv2d: y channels exceed stride");
  const int pad = (kw - 1) / 2;
  const int Wout = (Win + 2 * pad - kw) / stride_w + 1;  // mx Convolution output size
  RD_REQUIRE(!(flags & RD_SCALE_FOLDED), RD_EINVAL, "conv2d: RD_SCALE_FOLDED is taken by rd_conv3x3_bn_act_ex / rd_conv2d_bn_act_head_out / rd_deconv2d_bn_act");
  TapList tl = conv_taps(kh, kw);
  allow_conv_lds();
  return launch_conv(tl, x, x_cstride, x_coff, w_packed, scale, shift, residual, r_cstride, r_coff, y, y_cstride,
                     y_coff, B, H, Win, Wout, Wout, cin, cout, stride_w, 1, 0, flags, dtype, (hipStream_t)stream);
}
// ---- 3x3 conv of a BasicBlock, extended forms (bf16, persistent kernel) -----------------------------------------------------
// stride (1,2) on the pixel-pair view: x [H][W][cs] with W even IS [H][W/2][2*cs] -- channels [0, cs) of view pixel w2 are
// pixel 2*w2, channels [cs, 2cs) pixel 2*w2+1 -- and  out[w2] = sum_dh ( W[dh,-1].odd[w2-1] + W[dh,0].even[w2] + W[dh,+1].odd[w2] )
// is a stride-1 conv over the view with the six taps dw2 in {-1, 0} (tap set 1): two thirds of the multiply-adds of the
// stride-1-then-drop form and only the stored pixels are computed.
static int ex_view_cin(int cin, int x_cstride, int stride_w) { return stride_w == 2 ? x_cstride + cin : cin; }
size_t rd_conv3x3_ex_packed_bytes(int cin, int cout, int stride_w, int x_cstride) {
  return conv_packed_bytes(stride_w == 2 ? 6 : 9, ex_view_cin(cin, x_cstride, stride_w), cout, RD_BF16);
}
int rd_pack_conv3x3_ex_host(const float* w, const float* fold_scale, int cout, int cin, int stride_w, int x_cstride, int dtype,
                            void* out) {
  RD_REQUIRE(w && out, RD_EINVAL, "pack_conv3x3_ex: null pointer");
  RD_REQUIRE(is_h16(dtype), RD_EINVAL, "pack_conv3x3_ex: dtype %d (RD_BF16 or RD_F16)", dtype);
  RD_REQUIRE(stride_w == 1 || stride_w == 2, RD_ESHAPE, "pack_conv3x3_ex: stride_w %d", stride_w);
  RD_REQUIRE(cout == 64 || cout == 128, RD_ESHAPE, "pack_conv3x3_ex: cout %d", cout);
  RD_REQUIRE(stride_w == 1 || (x_cstride >= cin && x_cstride % 8 == 0), RD_ESHAPE, "pack_conv3x3_ex: x_cstride %d for cin %d", x_cstride, cin);
  auto wv = [&](int co, int ci, int dh, int dw) {
    return (fold_scale ? fold_scale[co] : 1.f) * w[(((size_t)co * cin + ci) * 3 + (dh + 1)) * 3 + (dw + 1)];
  };
  memset(out, 0, rd_conv3x3_ex_packed_bytes(cin, cout, stride_w, x_cstride));
  // pair view (stride 2): channel c2 < cin is the even pixel (only dw2 = 0 meets it), c2 >= x_cstride the odd one (dw = -1 / +1)
  auto wv2 = [&](int co, int c2, int dh, int dw2) -> float {
    if (c2 < cin) return dw2 == 0 ? wv(co, c2, dh, 0) : 0.f;
    if (c2 >= x_cstride && c2 - x_cstride < cin) return wv(co, c2 - x_cstride, dh, dw2 == -1 ? -1 : 1);
    return 0.f;
  };
  if (stride_w == 1) {
    if (conv3_body_small(cin, fold_scale != nullptr))   // <= 16 input channels: five two-tap steps instead of nine (k_conv3.h c3_body 3)
      pack_body_frag(3, cin, cout, out, [&](int co, int ci, int dh, int dw) { return wv(co, ci, dh - 1, dw - 1); }, dtype);
    else
      pack_taps_frag(9, cin, cout, out, [&](int co, int ci, int t) { return wv(co, ci, t / 3 - 1, t % 3 - 1); }, dtype);
  } else if (conv3_body_s2(cin, x_cstride, fold_scale != nullptr)) {
    // the even pixel's chunks as 3-tap units (no MFMAs on their structurally-zero dw2 = -1 weights): k_conv3.h c3_body 1
    pack_body_frag(1, x_cstride + cin, cout, out, [&](int co, int c2, int dh, int dw) { return wv2(co, c2, dh - 1, dw - 1); }, dtype);
  } else {
    pack_taps_frag(6, x_cstride + cin, cout, out, [&](int co, int c2, int s) -> float { return wv2(co, c2, s / 2 - 1, s % 2 - 1); }, dtype);
  }
  return RD_OK;
}
// the same conv for RD_MFMA16 launches (k_conv3.h M16: v_mfma_f32_16x16x32 fragments): stride 1, cout 128, cin a multiple of 32
int rd_pack_conv3x3_m16_host(const float* w, const float* fold_scale, int cout, int cin, int dtype, void* out) {
  RD_REQUIRE(w && out, RD_EINVAL, "pack_conv3x3_m16: null pointer");
  RD_REQUIRE(is_h16(dtype), RD_EINVAL, "pack_conv3x3_m16: dtype %d (RD_BF16 or RD_F16)", dtype);
  RD_REQUIRE(cout == 128 && cin >= 32 && cin % 32 == 0, RD_ESHAPE, "pack_conv3x3_m16: cout %d (128), cin %d (a multiple of 32)", cout, cin);
  memset(out, 0, rd_conv3x3_ex_packed_bytes(cin, cout, 1, cin));
  pack_taps_frag16(9, cin, cout, out, [&](int co, int ci, int t) {
    return (fold_scale ? fold_scale[co] : 1.f) * w[(((size_t)co * cin + ci) * 3 + t / 3) * 3 + t % 3];
  }, dtype);
  return RD_OK;
}
int rd_conv3x3_mfma16_ok(int cin, int cout, int stride_w, int W, int fused_output_conv) {
  return conv3_mfma16_ok(cin, cout, stride_w, W, fused_output_conv != 0) ? 1 : 0;
}
size_t rd_conv1x1_sc_packed_bytes(int cin, int cout) { return sc_frag_bytes(cin, cout); }
int rd_pack_conv1x1_sc_host(const float* w, const float* fold_scale, int cout, int cin, int dtype, void* out) {
  RD_REQUIRE(w && out, RD_EINVAL, "pack_conv1x1_sc: null pointer");
  RD_REQUIRE(is_h16(dtype), RD_EINVAL, "pack_conv1x1_sc: dtype %d (RD_BF16 or RD_F16)", dtype);
  RD_REQUIRE((cout == 64 || cout == 128) && cin >= 1 && cin <= 128, RD_ESHAPE, "pack_conv1x1_sc: cout %d, cin %d (1..128)", cout, cin);
  pack_sc_frag(w, fold_scale, cin, cout, out, dtype);
  return RD_OK;
}
int rd_conv3x3_bn_act_ex(const void* x, int x_cstride, int x_coff, const void* w_packed, const float* scale, const float* shift,
                         const void* residual, int r_cstride, int r_coff, const void* sc_x, int sc_cstride, int sc_coff,
                         int sc_cin, const void* sc_w_packed, void* y, int y_cstride, int y_coff, int B, int H, int Win, int cin,
                         int cout, int stride_w, int flags, int dtype, void* stream) {
  RD_REQUIRE(x && w_packed && y, RD_EINVAL, "conv3x3_ex: null pointer");
  RD_REQUIRE(is_h16(dtype), RD_EINVAL, "conv3x3_ex: dtype %d (RD_BF16 or RD_F16)", dtype);
  RD_REQUIRE(B > 0 && H > 0 && Win > 0 && cin > 0 && (cout == 64 || cout == 128), RD_ESHAPE, "conv3x3_ex: shape / cout %d", cout);
  RD_REQUIRE(stride_w == 1 || (stride_w == 2 && Win % 2 == 0), RD_ESHAPE, "conv3x3_ex: stride_w %d with Win %d (stride 2 needs an even width)", stride_w, Win);
  RD_REQUIRE(y_coff >= 0 && y_coff + cout <= y_cstride, RD_ESHAPE, "conv3x3_ex: y channels exceed stride");
  RD_REQUIRE(x_cstride % 8 == 0 && x_coff % 8 == 0 && x_coff + cin_slots(cin, RD_BF16) * 8 <= x_cstride, RD_ESHAPE, "conv3x3_ex: x channel stride/offset");
  RD_REQUIRE(!(sc_x && residual) && !(sc_x && !sc_w_packed), RD_EINVAL, "conv3x3_ex: shortcut conv and residual are exclusive");
  RD_REQUIRE(!((flags & RD_ADD) && !residual && !sc_x), RD_EINVAL, "conv3x3_ex: RD_ADD without residual / shortcut");
  RD_REQUIRE(!dev_switches().conv_v1, RD_EINVAL, "conv3x3_ex: needs the persistent 3x3 kernel (RD_CONV_V1 is set)");
  allow_conv_lds();
  const int v = stride_w;                      // pixels per view pixel
  const int Wv = Win / v;
  Conv3Args e;
  memset(&e, 0, sizeof(e));
  int fl = flags;
  if (sc_x) {
    RD_REQUIRE(sc_cin >= 1 && sc_cin <= 128 && sc_cstride % 8 == 0 && sc_coff % 8 == 0 && sc_coff + ((sc_cin + 15) / 16) * 16 <= sc_cstride,
               RD_ESHAPE, "conv3x3_ex: shortcut input channels %d (stride %d, offset %d)", sc_cin, sc_cstride, sc_coff);
    // the shortcut reads the block input at the OUTPUT pixel grid: stride 2 = the even pixel of each pair of the view
    e.sx = (const bf16_t*)sc_x; e.s_cs = sc_cstride * v; e.s_co = sc_coff; e.s_bs = (long)H * Wv * sc_cstride * v;
    e.scw = (const unsigned char*)sc_w_packed; e.s_nks = (sc_cin + 15) / 16;
    fl &= ~RD_ADD;                             // the add happens on the accumulators
    fl |= RD_SCALE_FOLDED;                     // (both scales are in the two weight sets)
  }
  RD_REQUIRE(!(fl & RD_SCALE_FOLDED) || !scale, RD_EINVAL, "conv3x3_ex: folded weights take no scale array");
  const bool m16 = (fl & RD_MFMA16) != 0;
  fl &= ~RD_MFMA16;
  const bool folded = (fl & RD_SCALE_FOLDED) != 0;
  RD_REQUIRE(!m16 || (folded && !sc_x && conv3_mfma16_ok(cin, cout, stride_w, Win, false)), RD_ESHAPE,
             "conv3x3_ex: RD_MFMA16 takes folded weights of rd_pack_conv3x3_m16_host: stride 1, cout 128, cin a multiple of 32, no fused shortcut (rd_conv3x3_mfma16_ok)");
  const int body = m16 ? C3_BODY_M16 : stride_w == 2 ? conv3_body_s2(cin, x_cstride, folded) : (sc_x ? 0 : conv3_body_small(cin, folded));
  RD_REQUIRE(!(stride_w == 1 && sc_x && conv3_body_small(cin, folded)), RD_ESHAPE, "conv3x3_ex: at most 16 input channels together with a fused shortcut is not a launch form");
  return launch_conv3(x, x_cstride * v, x_coff, w_packed, scale, shift, residual, r_cstride, r_coff, y, y_cstride, y_coff, B, H,
                      Wv, ex_view_cin(cin, x_cstride, stride_w), cout, fl, 1, (hipStream_t)stream, stride_w == 2 ? 1 : 0,
                      sc_x ? &e : nullptr, dtype, nullptr, nullptr, nullptr, body);
}

// ---- a whole 64-channel BasicBlock in one launch (k_block.h) ------------------------------------------------------------------------
size_t rd_block64_packed_bytes(int cin) { return block64_packed_bytes(cin); }
int rd_pack_block64_host(const float* w1, const float* fold_scale1, const float* w2, const float* fold_scale2, int cin, int dtype, void* out) {
  RD_REQUIRE(w1 && w2 && out, RD_EINVAL, "pack_block64: null pointer");
  RD_REQUIRE(is_h16(dtype), RD_EINVAL, "pack_block64: dtype %d (RD_BF16 or RD_F16)", dtype);
  RD_REQUIRE(cin == 64 || (cin >= 1 && cin <= 16), RD_ESHAPE, "pack_block64: %d input channels (64, or at most 16)", cin);
  pack_block64(w1, fold_scale1, w2, fold_scale2, cin, dtype, out);
  return RD_OK;
}
int rd_block64_bn_act(const void* x, int x_cstride, int x_coff, int cin, const void* w_packed, const float* shift1, const float* shift2,
                      const void* sc_w_packed, void* y, int y_cstride, int y_coff, int B, int H, int W, int dtype, void* stream) {
  RD_REQUIRE(x && w_packed && shift1 && shift2 && y, RD_EINVAL, "block64: null pointer");
  RD_REQUIRE(is_h16(dtype), RD_EINVAL, "block64: dtype %d (RD_BF16 or RD_F16)", dtype);
  RD_REQUIRE(B > 0 && H > 0 && W > 0, RD_ESHAPE, "block64: shape");
  RD_REQUIRE(cin == 64 || (cin >= 1 && cin <= 16), RD_ESHAPE, "block64: %d input channels (64, or at most 16 for the network's first block)", cin);
  RD_REQUIRE(x_cstride % 8 == 0 && x_coff % 8 == 0 && x_coff >= 0 && x_coff + cin_slots(cin, RD_BF16) * 8 <= x_cstride, RD_ESHAPE, "block64: x channel stride / offset");
  // (the first block's shortcut epilogue loads a whole 16-channel k-slot of x per pixel -- cin_slots() rounds cin <= 16 up to
  //  exactly that, so an 8-channel pitch is refused above; said once more explicitly so that it survives a change of cin_slots)
  RD_REQUIRE(cin > 16 || x_coff + 16 <= x_cstride, RD_ESHAPE, "block64: the first block reads 16 channels per pixel (x_coff %d + 16 > x_cstride %d)", x_coff, x_cstride);
  RD_REQUIRE(y_cstride % 8 == 0 && y_coff % 8 == 0 && y_coff >= 0 && y_coff + 64 <= y_cstride, RD_ESHAPE, "block64: y channel stride / offset");
  RD_REQUIRE(x != y, RD_EINVAL, "block64: in-place (a tile's halo is another tile's output)");
  return launch_block64(x, x_cstride, x_coff, cin, w_packed, shift1, shift2, sc_w_packed, y, y_cstride, y_coff, B, H, W, dtype, (hipStream_t)stream);
}

// ... in the 16 x 16 x 32 MFMA form (k_block.h M16; 64 input channels): its own weight images, the same results
int rd_pack_block64_m16_host(const float* w1, const float* fold_scale1, const float* w2, const float* fold_scale2, int dtype, void* out) {
  RD_REQUIRE(w1 && w2 && out, RD_EINVAL, "pack_block64_m16: null pointer");
  RD_REQUIRE(is_h16(dtype), RD_EINVAL, "pack_block64_m16: dtype %d (RD_BF16 or RD_F16)", dtype);
  pack_block64_m16(w1, fold_scale1, w2, fold_scale2, dtype, out);
  return RD_OK;
}
int rd_pack_conv1x1_sc_m16_host(const float* w, const float* fold_scale, int dtype, void* out) {
  RD_REQUIRE(w && out, RD_EINVAL, "pack_conv1x1_sc_m16: null pointer");
  RD_REQUIRE(is_h16(dtype), RD_EINVAL, "pack_conv1x1_sc_m16: dtype %d (RD_BF16 or RD_F16)", dtype);
  pack_sc_frag16(w, fold_scale, out, dtype);
  return RD_OK;
}
int rd_block64_m16_bn_act(const void* x, int x_cstride, int x_coff, const void* w_packed, const float* shift1, const float* shift2,
                          const void* sc_w_packed, void* y, int y_cstride, int y_coff, int B, int H, int W, int dtype, void* stream) {
  RD_REQUIRE(x && w_packed && shift1 && shift2 && y, RD_EINVAL, "block64_m16: null pointer");
  RD_REQUIRE(is_h16(dtype), RD_EINVAL, "block64_m16: dtype %d (RD_BF16 or RD_F16)", dtype);
  RD_REQUIRE(B > 0 && H > 0 && W > 0, RD_ESHAPE, "block64_m16: shape");
  RD_REQUIRE(x_cstride % 8 == 0 && x_coff % 8 == 0 && x_coff >= 0 && x_coff + 64 <= x_cstride, RD_ESHAPE, "block64_m16: x channel stride / offset");
  RD_REQUIRE(y_cstride % 8 == 0 && y_coff % 8 == 0 && y_coff >= 0 && y_coff + 64 <= y_cstride, RD_ESHAPE, "block64_m16: y channel stride / offset");
  RD_REQUIRE(x != y, RD_EINVAL, "block64_m16: in-place (a tile's halo is another tile's output)");
  return launch_block64(x, x_cstride, x_coff, 64, w_packed, shift1, shift2, sc_w_packed, y, y_cstride, y_coff, B, H, W, dtype, (hipStream_t)stream, true);
}

// ---- 3x3 conv over the channel concatenation [x1 | x2] of two tensors that is never materialised (16-bit, folded scales) -----------
// dla_backbone.py:153-154 (concat of the range image with the agg3 feature map) feeding head/builder.py:221-240: w_packed is
// rd_pack_conv3x3_cat_host of the weight whose input channels are laid out [cin1 channels of x1 | cin2 channels of x2].
size_t rd_conv3x3_cat_packed_bytes(int cin1, int cin2, int cout) { return conv_packed_bytes(9, cin1 + cin2, cout, RD_BF16); }
int rd_pack_conv3x3_cat_host(const float* w, const float* fold_scale, int cout, int cin1, int cin2, int dtype, void* out) {
  RD_REQUIRE(w && out, RD_EINVAL, "pack_conv3x3_cat: null pointer");
  RD_REQUIRE(is_h16(dtype), RD_EINVAL, "pack_conv3x3_cat: dtype %d (RD_BF16 or RD_F16)", dtype);
  RD_REQUIRE((cout == 64 || cout == 128) && cin1 > 0 && cin1 % 32 == 0 && cin2 > 0 && cin2 % 8 == 0, RD_ESHAPE,
             "pack_conv3x3_cat: cout %d, cin1 %d (multiple of 32), cin2 %d (multiple of 8)", cout, cin1, cin2);
  const int cin = cin1 + cin2;
  auto wv = [&](int co, int ci, int dh, int dw) {
    return (fold_scale ? fold_scale[co] : 1.f) * w[(((size_t)co * cin + ci) * 3 + (dh + 1)) * 3 + (dw + 1)];
  };
  memset(out, 0, rd_conv3x3_cat_packed_bytes(cin1, cin2, cout));
  const int body = conv3_body_cat(cin1, cin2, fold_scale != nullptr);
  if (body) pack_body_frag(body, cin, cout, out, [&](int co, int ci, int dh, int dw) { return wv(co, ci, dh - 1, dw - 1); }, dtype);
  else pack_taps_frag(9, cin, cout, out, [&](int co, int ci, int t) { return wv(co, ci, t / 3 - 1, t % 3 - 1); }, dtype);
  return RD_OK;
}
int rd_conv3x3_bn_act_cat(const void* x1, int x1_cstride, int x1_coff, int cin1, const void* x2, int x2_cstride, int x2_coff, int cin2,
                          const void* w_packed, const float* shift, void* y, int y_cstride, int y_coff, int B, int H, int W, int cout,
                          int flags, int dtype, void* stream) {
  RD_REQUIRE(x1 && x2 && w_packed && y && shift, RD_EINVAL, "conv3x3_cat: null pointer");
  RD_REQUIRE(is_h16(dtype), RD_EINVAL, "conv3x3_cat: dtype %d (RD_BF16 or RD_F16)", dtype);
  RD_REQUIRE((flags & RD_SCALE_FOLDED) && !(flags & RD_ADD), RD_EINVAL, "conv3x3_cat: needs RD_SCALE_FOLDED weights, takes no residual");
  RD_REQUIRE(B > 0 && H > 0 && W > 0 && (cout == 64 || cout == 128), RD_ESHAPE, "conv3x3_cat: shape / cout %d", cout);
  RD_REQUIRE(cin1 > 0 && cin1 % 32 == 0 && cin2 > 0 && cin2 % 8 == 0, RD_ESHAPE, "conv3x3_cat: cin1 %d (multiple of 32), cin2 %d (multiple of 8)", cin1, cin2);
  RD_REQUIRE(y_coff >= 0 && y_coff + cout <= y_cstride, RD_ESHAPE, "conv3x3_cat: y channels exceed stride");
  RD_REQUIRE(x1_cstride % 8 == 0 && x1_coff % 8 == 0 && x1_coff + cin1 <= x1_cstride && x2_cstride % 8 == 0 && x2_coff % 8 == 0 &&
             x2_coff + cin2 <= x2_cstride, RD_ESHAPE, "conv3x3_cat: channel stride/offset");
  const DevSwitches& sw_ = dev_switches();
  RD_REQUIRE(!sw_.conv_v1 && sw_.conv_wide && sw_.conv_w30 && (cout == 64 ? sw_.conv_th4 != 3 : (sw_.conv_th4 && sw_.conv_w30 == 2)), RD_ESHAPE,
             "conv3x3_cat: needs the 8 x 32 tile form of the persistent 3x3 kernel (a dev switch turned it off)");
  allow_conv_lds();
  Conv3Src2 s2;
  s2.x = x2; s2.cs = x2_cstride; s2.co = x2_coff; s2.cin1 = cin1; s2.cin2 = cin2;
  return launch_conv3(x1, x1_cstride, x1_coff, w_packed, nullptr, shift, nullptr, 0, 0, y, y_cstride, y_coff, B, H, W, cin1 + cin2, cout, flags,
                      1, (hipStream_t)stream, 0, nullptr, dtype, nullptr, nullptr, &s2, conv3_body_cat(cin1, cin2, true));
}

// ---- last tower conv + the tower's 1x1 output conv in one launch (bf16) ---------------------------------------------
size_t rd_head_packed_bytes(void) { return 16384; }
int rd_pack_head_weight_host(const float* w, int nout, int cin, int dtype, void* out_host) {
  RD_REQUIRE(w && out_host, RD_EINVAL, "pack_head_weight: null pointer");
  RD_REQUIRE(is_h16(dtype), RD_EINVAL, "pack_head_weight: dtype %d (RD_BF16 or RD_F16)", dtype);
  RD_REQUIRE(nout >= 1 && nout <= 8 && cin >= 1 && cin <= 128, RD_ESHAPE, "pack_head_weight: nout %d (1..8), cin %d (1..128)", nout, cin);
  pack_head_frag(w, nout, cin, out_host, dtype);
  return RD_OK;
}
// ... for RD_MFMA16 launches (k_conv3.h pack_head_frag16): 8 KB
size_t rd_head_m16_packed_bytes(void) { return 8192; }
int rd_pack_head_weight_m16_host(const float* w, int nout, int cin, int dtype, void* out_host) {
  RD_REQUIRE(w && out_host, RD_EINVAL, "pack_head_weight_m16: null pointer");
  RD_REQUIRE(is_h16(dtype), RD_EINVAL, "pack_head_weight_m16: dtype %d (RD_BF16 or RD_F16)", dtype);
  RD_REQUIRE(nout >= 1 && nout <= 8 && cin >= 1 && cin <= 128, RD_ESHAPE, "pack_head_weight_m16: nout %d (1..8), cin %d (1..128)", nout, cin);
  pack_head_frag16(w, nout, cin, out_host, dtype);
  return RD_OK;
}
int rd_conv2d_bn_act_head_out(const void* x, int x_cstride, int x_coff, const void* w_packed, const float* scale,
                              const float* shift, int B, int H, int W, int cin, int flags, const void* head_w_packed,
                              const float* head_bias, float* out, long out_batch_stride, long n_off, int nout, int dtype,
                              void* stream) {
  RD_REQUIRE(x && w_packed && head_w_packed && head_bias && out, RD_EINVAL, "conv2d_head_out: null pointer");
  RD_REQUIRE(is_h16(dtype), RD_EINVAL, "conv2d_head_out: dtype %d (RD_BF16 or RD_F16)", dtype);
  RD_REQUIRE(B > 0 && H > 0 && W > 0 && cin > 0 && nout >= 1 && nout <= 8, RD_ESHAPE, "conv2d_head_out: shape / nout %d (1..8)", nout);
  RD_REQUIRE(!(flags & RD_ADD), RD_EINVAL, "conv2d_head_out: no residual in a head tower");
  RD_REQUIRE(x_cstride % 8 == 0 && x_coff % 8 == 0 && x_coff + cin_slots(cin, RD_BF16) * 8 <= x_cstride, RD_ESHAPE,
             "conv2d_head_out: x channel stride/offset");
  RD_REQUIRE(!dev_switches().conv_v1, RD_EINVAL, "conv2d_head_out: needs the persistent 3x3 kernel (RD_CONV_V1 is set)");
  RD_REQUIRE(!conv3_body_small(cin, (flags & RD_SCALE_FOLDED) != 0), RD_ESHAPE, "conv2d_head_out: cin %d (the packed image of a conv with <= 16 input channels is not this launch form's)", cin);
  const bool m16 = (flags & RD_MFMA16) != 0;
  RD_REQUIRE(!m16 || ((flags & RD_SCALE_FOLDED) && conv3_mfma16_ok(cin, 128, 1, W, true)), RD_ESHAPE,
             "conv2d_head_out: RD_MFMA16 (weights of rd_pack_conv3x3_m16_host) is not a launch form for cin %d at width %d (rd_conv3x3_mfma16_ok)", cin, W);
  allow_conv_lds();
  Conv3Args h;
  memset(&h, 0, sizeof(h));
  h.hw = (const unsigned char*)head_w_packed; h.hb = head_bias; h.ho = out; h.ho_bs = out_batch_stride; h.ho_off = n_off; h.hn = nout;
  return launch_conv3(x, x_cstride, x_coff, w_packed, scale, shift, nullptr, 0, 0, nullptr, 128, 0, B, H, W, cin, 128, flags & ~RD_MFMA16, 1,
                      (hipStream_t)stream, 0, &h, dtype, nullptr, nullptr, nullptr, m16 ? C3_BODY_M16 : 0);
}

// ---- the cls and the reg tower conv of a head level as ONE launch (two problems of the same shape) -------------------------
static int pair_checks(const char* fn, const void* x0, const void* x1, const void* w0, const void* w1, const float* shift0,
                       const float* shift1, int x_cstride, int x0_coff, int x1_coff, int B, int H, int W, int cin, int flags,
                       int dtype) {
  RD_REQUIRE(x0 && x1 && w0 && w1 && shift0 && shift1, RD_EINVAL, "%s: null pointer", fn);
  RD_REQUIRE(is_h16(dtype), RD_EINVAL, "%s: dtype %d (RD_BF16 or RD_F16)", fn, dtype);
  RD_REQUIRE(B > 0 && H > 0 && W > 0 && cin > 0, RD_ESHAPE, "%s: shape", fn);
  RD_REQUIRE((flags & RD_SCALE_FOLDED) && !(flags & RD_ADD), RD_EINVAL, "%s: needs RD_SCALE_FOLDED weights, takes no residual", fn);
  RD_REQUIRE(x_cstride % 8 == 0 && x0_coff % 8 == 0 && x1_coff % 8 == 0 && x0_coff + cin_slots(cin, RD_BF16) * 8 <= x_cstride &&
             x1_coff + cin_slots(cin, RD_BF16) * 8 <= x_cstride, RD_ESHAPE, "%s: x channel stride/offset", fn);
  RD_REQUIRE(!dev_switches().conv_v1, RD_EINVAL, "%s: needs the persistent 3x3 kernel (RD_CONV_V1 is set)", fn);
  RD_REQUIRE(cin > 16, RD_ESHAPE, "%s: cin %d (more than 16 input channels)", fn, cin);
  return RD_OK;
}
int rd_conv3x3_bn_act_pair(const void* x0, int x0_coff, const void* w0_packed, const float* shift0, void* y0, int y0_coff,
                           const void* x1, int x1_coff, const void* w1_packed, const float* shift1, void* y1, int y1_coff,
                           int x_cstride, int y_cstride, int B, int H, int W, int cin, int flags, int dtype, void* stream) {
  if (int rc = pair_checks("conv3x3_pair", x0, x1, w0_packed, w1_packed, shift0, shift1, x_cstride, x0_coff, x1_coff, B, H, W, cin, flags, dtype)) return rc;
  RD_REQUIRE(y0 && y1, RD_EINVAL, "conv3x3_pair: null output");
  RD_REQUIRE(y0_coff >= 0 && y1_coff >= 0 && y0_coff + 128 <= y_cstride && y1_coff + 128 <= y_cstride, RD_ESHAPE, "conv3x3_pair: y channels exceed stride");
  allow_conv_lds();
  if (!conv3_pair_eligible(128, flags, W)) {   // (a dev-switch combination without the two-workgroup 8 x 32 tiles: two launches)
    if (int rc = launch_conv3(x0, x_cstride, x0_coff, w0_packed, nullptr, shift0, nullptr, 0, 0, y0, y_cstride, y0_coff, B, H, W, cin, 128,
                              flags, 1, (hipStream_t)stream, 0, nullptr, dtype)) return rc;
    return launch_conv3(x1, x_cstride, x1_coff, w1_packed, nullptr, shift1, nullptr, 0, 0, y1, y_cstride, y1_coff, B, H, W, cin, 128, flags, 1,
                        (hipStream_t)stream, 0, nullptr, dtype);
  }
  Conv3Second g;
  memset(&g, 0, sizeof(g));
  g.x = x1; g.x_co = x1_coff; g.w = w1_packed; g.shift = shift1; g.y = y1; g.y_co = y1_coff;
  return launch_conv3(x0, x_cstride, x0_coff, w0_packed, nullptr, shift0, nullptr, 0, 0, y0, y_cstride, y0_coff, B, H, W, cin, 128, flags, 1,
                      (hipStream_t)stream, 0, nullptr, dtype, &g);
}
int rd_conv2d_bn_act_head_out_pair(const void* x0, int x0_coff, const void* w0_packed, const float* shift0, const void* head_w0_packed,
                                   const float* head_bias0, float* out0, long out0_batch_stride, int nout0,
                                   const void* x1, int x1_coff, const void* w1_packed, const float* shift1, const void* head_w1_packed,
                                   const float* head_bias1, float* out1, long out1_batch_stride, int nout1,
                                   int x_cstride, long n_off, int B, int H, int W, int cin, int flags, int dtype, void* stream) {
  if (int rc = pair_checks("conv2d_head_out_pair", x0, x1, w0_packed, w1_packed, shift0, shift1, x_cstride, x0_coff, x1_coff, B, H, W, cin, flags, dtype)) return rc;
  RD_REQUIRE(head_w0_packed && head_w1_packed && head_bias0 && head_bias1 && out0 && out1, RD_EINVAL, "conv2d_head_out_pair: null pointer");
  RD_REQUIRE(nout0 >= 1 && nout0 <= 8 && nout1 >= 1 && nout1 <= 8, RD_ESHAPE, "conv2d_head_out_pair: nout %d / %d (1..8)", nout0, nout1);
  allow_conv_lds();
  Conv3Args h;
  memset(&h, 0, sizeof(h));
  h.hw = (const unsigned char*)head_w0_packed; h.hb = head_bias0; h.ho = out0; h.ho_bs = out0_batch_stride; h.ho_off = n_off; h.hn = nout0;
  if (!conv3_pair_eligible(128, flags, W, true)) {
    if (int rc = launch_conv3(x0, x_cstride, x0_coff, w0_packed, nullptr, shift0, nullptr, 0, 0, nullptr, 128, 0, B, H, W, cin, 128, flags, 1,
                              (hipStream_t)stream, 0, &h, dtype)) return rc;
    h.hw = (const unsigned char*)head_w1_packed; h.hb = head_bias1; h.ho = out1; h.ho_bs = out1_batch_stride; h.hn = nout1;
    return launch_conv3(x1, x_cstride, x1_coff, w1_packed, nullptr, shift1, nullptr, 0, 0, nullptr, 128, 0, B, H, W, cin, 128, flags, 1,
                        (hipStream_t)stream, 0, &h, dtype);
  }
  Conv3Second g;
  memset(&g, 0, sizeof(g));
  g.x = x1; g.x_co = x1_coff; g.w = w1_packed; g.shift = shift1;
  g.hw = head_w1_packed; g.hb = head_bias1; g.ho = out1; g.ho_bs = out1_batch_stride; g.hn = nout1;
  return launch_conv3(x0, x_cstride, x0_coff, w0_packed, nullptr, shift0, nullptr, 0, 0, nullptr, 128, 0, B, H, W, cin, 128, flags, 1,
                      (hipStream_t)stream, 0, &h, dtype, &g);
}

int rd_deconv2d_bn_act(const void* x, int x_cstride, int x_coff, const void* w_packed_phase, const float* scale,
                       const float* shift, const void* residual, int r_cstride, int r_coff, void* y, int y_cstride,
                       int y_coff, int B, int H, int Win, int cin, int cout, int kh, int kw, int stride_w, int pad_w,
                       int phase, int flags, int dtype, void* stream) {
  RD_REQUIRE(x && w_packed_phase && y, RD_EINVAL, "deconv2d: null pointer");
  RD_REQUIRE(kh == 3, RD_ESHAPE, "deconv2d: kernel height %d (only 3, pad 1, stride 1)", kh);
  RD_REQUIRE(stride_w >= 1 && phase >= 0 && phase < stride_w, RD_EINVAL, "deconv2d: phase %d of stride %d", phase, stride_w);
  RD_REQUIRE(y_coff >= 0 && y_coff + cout <= y_cstride, RD_ESHAPE, "deconv2d: y channels exceed stride");
  const int Wout = (Win - 1) * stride_w - 2 * pad_w + kw;  // mx Deconvolution output size
  RD_REQUIRE(Wout > phase, RD_ESHAPE, "deconv2d: empty phase");
  const int Wq = (Wout - phase + stride_w - 1) / stride_w;
  TapList tl = deconv_taps_sorted(kh, kw, stride_w, pad_w, phase);
  RD_REQUIRE(tl.n >= 1 && tl.n <= 9, RD_ESHAPE, "deconv2d: %d taps per phase unsupported", tl.n);
  allow_conv_lds();
  if (deconv_embeds_3x3(tl)) {
    const int ts = deconv_tap_set(tl);
    if (!ts) tl = conv_taps(3, 3);   // packed as a full 3x3 window with zero weights (rd_pack_deconv_weight_host)
    if (is_h16(dtype) && Wout == stride_w * Win && (cout == 64 || cout == 128) && !dev_switches().conv_v1 &&
        conv3_has_form(dtype, (flags & RD_SCALE_FOLDED) != 0)) {
      // phase pixels of the output seen as [H][Win][stride_w * Cstride]: channel offset phase * Cstride
      const bf16_t* r = (const bf16_t*)residual;
      return launch_conv3(x, x_cstride, x_coff, w_packed_phase, scale, shift, r, r_cstride * stride_w,
                          r_coff + phase * r_cstride, y, y_cstride * stride_w, y_coff + phase * y_cstride, B, H, Win, cin,
                          cout, flags, 1, (hipStream_t)stream, ts, nullptr, dtype);
    }
  }
  RD_REQUIRE(!(flags & RD_SCALE_FOLDED), RD_ESHAPE, "deconv2d: RD_SCALE_FOLDED needs the persistent 3x3 kernel (bf16, cout 64/128, "
             "taps inside the 3x3 window, Wout == stride_w * Win)");
  return launch_conv(tl, x, x_cstride, x_coff, w_packed_phase, scale, shift, residual, r_cstride, r_coff, y,
                     y_cstride, y_coff, B, H, Win, Wq, Wout, cin, cout, 1, stride_w, phase, flags, dtype,
                     (hipStream_t)stream);
}

// ---- ALL phases of a transposed conv in ONE launch (16-bit, folded scales) -------------------------------------------------------
// w_packed_all = the stride_w phase images of rd_pack_deconv_weight_folded_host one after the other, w_phase_bytes apart.
int rd_deconv2d_all_phases_ok(int kh, int kw, int stride_w, int pad_w, int cout, int dtype) {
  if (!is_h16(dtype) || kh != 3 || stride_w < 2 || stride_w > 8 || !conv3_phases_eligible(cout, RD_SCALE_FOLDED)) return 0;
  if (kw - 2 * pad_w != stride_w) return 0;   // Wout = (Win - 1) * stride_w - 2 * pad_w + kw must equal stride_w * Win
  for (int p = 0; p < stride_w; ++p)
    if (!deconv_tap_set(deconv_taps_sorted(kh, kw, stride_w, pad_w, p))) return 0;
  return 1;
}
int rd_deconv2d_bn_act_all(const void* x, int x_cstride, int x_coff, const void* w_packed_all, long w_phase_bytes,
                           const float* shift, const void* residual, int r_cstride, int r_coff, void* y, int y_cstride,
                           int y_coff, int B, int H, int Win, int cin, int cout, int kh, int kw, int stride_w, int pad_w,
                           int flags, int dtype, void* stream) {
  RD_REQUIRE(x && w_packed_all && y, RD_EINVAL, "deconv2d_all: null pointer");
  RD_REQUIRE(is_h16(dtype), RD_EINVAL, "deconv2d_all: dtype %d (RD_BF16 or RD_F16)", dtype);
  RD_REQUIRE(flags & RD_SCALE_FOLDED, RD_EINVAL, "deconv2d_all: needs RD_SCALE_FOLDED weights (rd_pack_deconv_weight_folded_host)");
  RD_REQUIRE(!((flags & RD_ADD) && !residual), RD_EINVAL, "deconv2d_all: RD_ADD without residual");
  RD_REQUIRE(B > 0 && H > 0 && Win > 0 && cin > 0, RD_ESHAPE, "deconv2d_all: shape");
  RD_REQUIRE(y_coff >= 0 && y_coff + cout <= y_cstride, RD_ESHAPE, "deconv2d_all: y channels exceed stride");
  RD_REQUIRE(x_cstride % 8 == 0 && x_coff % 8 == 0 && x_coff + cin_slots(cin, RD_BF16) * 8 <= x_cstride, RD_ESHAPE, "deconv2d_all: x channel stride/offset");
  RD_REQUIRE(rd_deconv2d_all_phases_ok(kh, kw, stride_w, pad_w, cout, dtype), RD_ESHAPE,
             "deconv2d_all: kernel (%d,%d) stride %d pad %d cout %d is not a set of two-column 3x2 phases (use rd_deconv2d_bn_act per phase)",
             kh, kw, stride_w, pad_w, cout);
  RD_REQUIRE(w_phase_bytes >= (long)conv_packed_bytes(6, cin, cout, RD_BF16) && w_phase_bytes % 16 == 0, RD_EINVAL, "deconv2d_all: w_phase_bytes");
  allow_conv_lds();
  Conv3Phases ph;
  ph.nph = stride_w; ph.ts_mask = 0; ph.y_pc = y_cstride; ph.r_pc = r_cstride; ph.w_pb = w_phase_bytes;
  for (int p = 0; p < stride_w; ++p)
    if (deconv_tap_set(deconv_taps_sorted(kh, kw, stride_w, pad_w, p)) == 2) ph.ts_mask |= 1 << p;
  return launch_conv3(x, x_cstride, x_coff, w_packed_all, nullptr, shift, residual, r_cstride * stride_w, r_coff, y, y_cstride * stride_w,
                      y_coff, B, H, Win, cin, cout, flags, 1, (hipStream_t)stream, 3, nullptr, dtype, nullptr, &ph);
}

// ---- phase PAIRS of a transposed conv (cout 64, stride 4: dla_backbone.py:117-127 agg1) ----------------------------------------
// Output phases 2p and 2p+1 of a (3, 2*s) / stride s / pad s/2 transposed conv with s = 4 read the SAME two input columns (tap set 1
// for the pair 0 | 1, tap set 2 for 2 | 3), and in the output seen as [H][Win][s * C] their channels are neighbours: a pair is ONE
// 3 x 2-tap conv with 2 * cout = 128 output channels.  It runs the cout-128 form of the persistent kernel (every pixel fragment read
// from LDS feeds four MFMAs instead of two, half as many tile passes and epilogues) with bit-identical results: the K order of every
// output element is unchanged.  Needs a dense cout-channel output / residual (y_cstride == r_cstride == cout, offsets 0).
int rd_deconv2d_phase_pairs_ok(int kh, int kw, int stride_w, int pad_w, int cout, int dtype) {
  if (!rd_deconv2d_all_phases_ok(kh, kw, stride_w, pad_w, cout, dtype) || cout != 64 || stride_w % 2) return 0;
  if (!rd_deconv2d_all_phases_ok(kh, kw, stride_w, pad_w, 128, dtype)) return 0;    // (the cout-128 form of the kernel is available)
  for (int p = 0; p < stride_w; p += 2)
    if (deconv_tap_set(deconv_taps_sorted(kh, kw, stride_w, pad_w, p)) != deconv_tap_set(deconv_taps_sorted(kh, kw, stride_w, pad_w, p + 1))) return 0;
  return 1;
}
// pair image = the two phase images of rd_pack_deconv_weight_folded_host interleaved per (chunk, tap, k-step): [A: 2 KB | B: 2 KB]
int rd_pack_deconv_phase_pair_host(const void* phase_a, const void* phase_b, int cin, int dtype, void* out) {
  RD_REQUIRE(phase_a && phase_b && out, RD_EINVAL, "pack_deconv_pair: null pointer");
  RD_REQUIRE(is_h16(dtype) && cin > 0, RD_EINVAL, "pack_deconv_pair: dtype %d / cin %d", dtype, cin);
  const size_t body = conv_packed_body_bytes(6, cin, 64, RD_BF16), blocks = body / 2048;    // (chunk, tap, k-step) blocks of 2 x 1 KB
  for (size_t i = 0; i < blocks; ++i) {
    memcpy((unsigned char*)out + i * 4096, (const unsigned char*)phase_a + i * 2048, 2048);
    memcpy((unsigned char*)out + i * 4096 + 2048, (const unsigned char*)phase_b + i * 2048, 2048);
  }
  memset((unsigned char*)out + 2 * body, 0, 2 * RD_CONV_TAIL);    // the zero tail the kernel's padding DMA reads (k_conv.h RD_CONV_TAIL)
  return RD_OK;
}
// w_packed_pairs: stride_w / 2 pair images, w_pair_bytes apart; shift2: 2 * cout values (the layer's shift twice)
int rd_deconv2d_bn_act_pairs(const void* x, int x_cstride, int x_coff, const void* w_packed_pairs, long w_pair_bytes,
                             const float* shift2, const void* residual, int r_cstride, int r_coff, void* y, int y_cstride,
                             int y_coff, int B, int H, int Win, int cin, int cout, int kh, int kw, int stride_w, int pad_w,
                             int flags, int dtype, void* stream) {
  RD_REQUIRE(x && w_packed_pairs && y && shift2, RD_EINVAL, "deconv2d_pairs: null pointer");
  RD_REQUIRE(is_h16(dtype), RD_EINVAL, "deconv2d_pairs: dtype %d (RD_BF16 or RD_F16)", dtype);
  RD_REQUIRE(flags & RD_SCALE_FOLDED, RD_EINVAL, "deconv2d_pairs: needs RD_SCALE_FOLDED weights (rd_pack_deconv_weight_folded_host)");
  RD_REQUIRE(!((flags & RD_ADD) && !residual), RD_EINVAL, "deconv2d_pairs: RD_ADD without residual");
  RD_REQUIRE(B > 0 && H > 0 && Win > 0 && cin > 0, RD_ESHAPE, "deconv2d_pairs: shape");
  RD_REQUIRE(x_cstride % 8 == 0 && x_coff % 8 == 0 && x_coff + cin_slots(cin, RD_BF16) * 8 <= x_cstride, RD_ESHAPE, "deconv2d_pairs: x channel stride/offset");
  RD_REQUIRE(rd_deconv2d_phase_pairs_ok(kh, kw, stride_w, pad_w, cout, dtype), RD_ESHAPE,
             "deconv2d_pairs: kernel (%d,%d) stride %d pad %d cout %d has no phase pairs (use rd_deconv2d_bn_act_all)", kh, kw, stride_w, pad_w, cout);
  RD_REQUIRE(y_cstride == cout && y_coff == 0 && (!residual || (r_cstride == cout && r_coff == 0)), RD_ESHAPE,
             "deconv2d_pairs: the output / residual must be dense %d-channel tensors (two phases are one 128-channel write)", cout);
  RD_REQUIRE(w_pair_bytes >= (long)conv_packed_bytes(6, cin, 2 * cout, RD_BF16) && w_pair_bytes % 16 == 0, RD_EINVAL, "deconv2d_pairs: w_pair_bytes");
  allow_conv_lds();
  Conv3Phases ph;
  ph.nph = stride_w / 2; ph.ts_mask = 0; ph.y_pc = 2 * cout; ph.r_pc = 2 * cout; ph.w_pb = w_pair_bytes;
  for (int p = 0; p < stride_w; p += 2)
    if (deconv_tap_set(deconv_taps_sorted(kh, kw, stride_w, pad_w, p)) == 2) ph.ts_mask |= 1 << (p / 2);
  return launch_conv3(x, x_cstride, x_coff, w_packed_pairs, nullptr, shift2, residual, r_cstride * stride_w, r_coff, y, y_cstride * stride_w,
                      y_coff, B, H, Win, cin, 2 * cout, flags, 1, (hipStream_t)stream, 3, nullptr, dtype, nullptr, &ph);
}

int rd_head_out(const void* x, int x_cstride, int x_coff, const float* w, const float* bias, float* out,
                long out_batch_stride, long n_off, int B, int H, int W, int cin, int nout, int dtype, void* stream) {
  RD_REQUIRE(x && w && bias && out, RD_EINVAL, "head_out: null pointer");
  RD_REQUIRE(cin % 8 == 0 && cin <= 128 && cin > 0, RD_ESHAPE, "head_out: cin %d (multiple of 8, <= 128)", cin);
  RD_REQUIRE(nout == 1 || nout == 7 || nout == 8, RD_ESHAPE, "head_out: nout %d not in {1,7,8}", nout);
  RD_REQUIRE(dtype == RD_F32 || is_h16(dtype), RD_EINVAL, "head_out: dtype");
  hipStream_t st = (hipStream_t)stream;
  const long HW = (long)H * W;
  dim3 grid((unsigned)std::min<long>((HW + 31) / 32, 4096), B);
  ProfScope ps(RD_PROF_HEAD_OUT, st);
  if (is_h16(dtype) && nout > 1 && cin % 16 == 0 && x_cstride % 8 == 0 && x_coff % 8 == 0) {   // matrix-core streaming variant
    dim3 g2((unsigned)std::min<long>((HW + 127) / 128, 2048), B);
#define RD_HOM(NO, DT_) hipLaunchKernelGGL((head_out_mfma_kernel<NO, DT_>), g2, dim3(256), 0, st, (const bf16_t*)x, x_cstride, x_coff, w, bias, out, out_batch_stride, n_off, HW, cin)
    if (dtype == RD_F16) { if (nout == 7) RD_HOM(7, RD_F16); else RD_HOM(8, RD_F16); }
    else { if (nout == 7) RD_HOM(7, RD_BF16); else RD_HOM(8, RD_BF16); }
#undef RD_HOM
    return check_launch("head_out");
  }
#define RD_HO(DT, NO) hipLaunchKernelGGL((head_out_kernel<DT, NO>), grid, dim3(256), 0, st, x, x_cstride, x_coff, w, bias, out, out_batch_stride, n_off, HW, cin)
  if (dtype == RD_BF16) { if (nout == 1) RD_HO(RD_BF16, 1); else if (nout == 7) RD_HO(RD_BF16, 7); else RD_HO(RD_BF16, 8); }
  else if (dtype == RD_F16) { if (nout == 1) RD_HO(RD_F16, 1); else if (nout == 7) RD_HO(RD_F16, 7); else RD_HO(RD_F16, 8); }
  else { if (nout == 1) RD_HO(RD_F32, 1); else if (nout == 7) RD_HO(RD_F32, 7); else RD_HO(RD_F32, 8); }
#undef RD_HO
  return check_launch("head_out");
}

// ---- Meta-Kernel ---------------------------------------------------------------------------------------------
size_t rd_meta_packed_bytes(int dtype) { return meta_layout(dtype).total; }
int rd_pack_meta_host(const float* w0, const float* b0, const float* w1, const float* b1, const float* s1,
                      const float* t1, const float* agg, const float* s2, const float* t2, int dtype, void* out) {
  RD_REQUIRE(w0 && b0 && w1 && b1 && s1 && t1 && agg && s2 && t2 && out, RD_EINVAL, "pack_meta: null pointer");
  RD_REQUIRE(dtype == RD_F32 || is_h16(dtype), RD_EINVAL, "pack_meta: dtype");
  pack_meta(w0, b0, w1, b1, s1, t1, agg, s2, t2, dtype, out);
  return RD_OK;
}
int rd_meta_kernel_fwd(const void* data, int d_cstride, int d_coff, const float* coord_nchw, const void* packed,
                       void* y, int y_cstride, int y_coff, int B, int H, int W, int dtype, void* stream) {
  RD_REQUIRE(data && coord_nchw && packed && y, RD_EINVAL, "meta_kernel: null pointer");
  RD_REQUIRE(B > 0 && H > 0 && W > 0, RD_ESHAPE, "meta_kernel: empty shape");
  RD_REQUIRE(dtype == RD_F32 || is_h16(dtype), RD_EINVAL, "meta_kernel: dtype");
  const int ch = ch_per_slot(dtype);
  RD_REQUIRE(d_cstride % ch == 0 && d_coff % ch == 0 && y_cstride % ch == 0 && y_coff % ch == 0, RD_ESHAPE,
             "meta_kernel: channel strides/offsets must be 16-byte multiples");
  RD_REQUIRE(d_coff + 64 <= d_cstride && y_coff + 64 <= y_cstride, RD_ESHAPE, "meta_kernel: needs 64 channels");
  hipStream_t st = (hipStream_t)stream;
  MetaArgs a;
  a.data = data; a.d_cs = d_cstride; a.d_co = d_coff; a.coord = coord_nchw; a.packed = (const unsigned char*)packed;
  a.y = y; a.y_cs = y_cstride; a.y_co = y_coff; a.B = B; a.H = H; a.W = W;
  constexpr int WAVES = 8;
  a.tiles_h = (H + WAVES - 1) / WAVES;
  a.tiles_w = (W + 31) / 32;
  a.ntiles = B * a.tiles_h * a.tiles_w;
  a.r0 = 0; a.m0 = a.m1 = a.m2 = 0;
  const size_t consts = 9 * 64 * 4 * 2 + 1024;
  ProfScope ps(RD_PROF_META, st);
  if (is_h16(dtype)) {
    const size_t lds = meta_layout(dtype).wbytes + consts + (size_t)(WAVES + 2) * 34 * 128 + 4096;
    static std::atomic<unsigned long long> seen{0};
    once_per_device(seen, [] { allow_big_lds(meta16_kernel<WAVES, RD_BF16>); allow_big_lds(meta16_kernel<WAVES, RD_F16>); });
    const dim3 mgrid(std::min(a.ntiles, conv_num_cus())), mblock(WAVES * 64);
    const int strips = a.tiles_w * B;
    a.r0 = dev_switches().conv_xcd && strips % 8 == 0 ? 8 : strips;   // (MetaArgs::r0: XCD-aware tile order; a permutation for any grid)
    RD_REQUIRE((unsigned long long)a.ntiles * (unsigned)std::max(a.r0, std::max(a.tiles_h, a.tiles_w)) < (1ull << 32), RD_ESHAPE,
               "meta_kernel: %d tiles exceed the range of the tile decode", a.ntiles);
    a.m0 = meta_magic(a.r0); a.m1 = meta_magic(a.tiles_h); a.m2 = meta_magic(a.tiles_w);
    // (form 8 of the kernel: per-image halo offsets are 32-bit, built from 24-bit multiplies)
    RD_REQUIRE((long)H * W < (1l << 23) && (long)H * W * d_cstride < (1l << 31) && W < (1 << 23) && d_cstride < (1 << 23), RD_ESHAPE,
               "meta_kernel: image of %d x %d pixels x %d channels exceeds the 32-bit halo offsets", H, W, d_cstride);
    if (dtype == RD_F16) hipLaunchKernelGGL((meta16_kernel<WAVES, RD_F16>), mgrid, mblock, lds, st, a);
    else hipLaunchKernelGGL((meta16_kernel<WAVES, RD_BF16>), mgrid, mblock, lds, st, a);
  } else {
    const size_t lds = consts + (size_t)(WAVES + 2) * 34 * 256;
    static std::atomic<unsigned long long> seen32{0};
    once_per_device(seen32, [] { allow_big_lds(meta_kernel<RD_F32, WAVES>); });
    hipLaunchKernelGGL((meta_kernel<RD_F32, WAVES>), dim3(std::min(a.ntiles, 512)), dim3(WAVES * 64), lds, st, a);
  }
  return check_launch("meta_kernel");
}

// ---- post-processing -------------------------------------------------------------------------------------------
size_t rd_sorted_foreground_workspace_bytes(long N, long k) { (void)k; return (sort_ws_bytes(N) + 256 + 255) & ~(size_t)255; }
int rd_sorted_foreground(const float* cls_score, const float* bbox_delta, const float* pc, const float* mask, int B,
                         long N, long k, int D, int apply_sigmoid, float* out_score, float* out_delta, float* out_pc,
                         int* out_idx, void* ws, size_t ws_bytes, void* stream) {
  RD_REQUIRE(cls_score && bbox_delta && pc && out_score && out_delta && out_pc && ws, RD_EINVAL, "sorted_foreground: null pointer");
  RD_REQUIRE(B > 0 && N > 0 && k > 0 && D > 0, RD_ESHAPE, "sorted_foreground: empty shape");
  RD_REQUIRE(N >= k, RD_ESHAPE, "sorted_foreground: N (%ld) < num_fgs (%ld)", N, k);  // get_sorted_foreground.py:65
  RD_REQUIRE(N < (1L << 31), RD_ESHAPE, "sorted_foreground: N too large");
  RD_REQUIRE(ws_bytes >= rd_sorted_foreground_workspace_bytes(N, k), RD_EWORKSPACE, "sorted_foreground: workspace %zu < %zu",
             ws_bytes, rd_sorted_foreground_workspace_bytes(N, k));
  hipStream_t st = (hipStream_t)stream;
  void* wsa = (void*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  SortWs s = sort_ws_carve(wsa, N);
  ProfScope ps(RD_PROF_SORT, st);
  // the B batch elements are independent sorts: side by side (blockIdx.y) when the workspace holds B problems,
  // otherwise one after the other in a single-problem workspace
  const size_t per = rd_sorted_foreground_workspace_bytes(N, k);
  const bool wide = ws_bytes >= per * (size_t)B;
  const long stride = wide ? (long)(per / 4) : 0;
  const int nb = wide ? B : 1;
  for (int b0 = 0; b0 < B; b0 += nb) {
    hipLaunchKernelGGL(sort_keygen_kernel, dim3((unsigned)((N + 255) / 256), nb), dim3(256), 0, st, cls_score + (size_t)b0 * N,
                       mask ? mask + (size_t)b0 * N : nullptr, N, apply_sigmoid, s.keysA, s.idxA, stride);
    unsigned *rk, *ri;
    int rc = radix_topk_pairs(s, N, k, st, nb, stride, &rk, &ri);
    if (rc != RD_OK) return rc;
    hipLaunchKernelGGL(sort_gather_kernel, dim3((unsigned)((k + 255) / 256), nb), dim3(256), 0, st, rk, ri, k, D,
                       bbox_delta + (size_t)b0 * N * D, pc + (size_t)b0 * N * 3, out_score + (size_t)b0 * k,
                       out_delta + (size_t)b0 * k * D, out_pc + (size_t)b0 * k * 3, out_idx ? out_idx + (size_t)b0 * k : nullptr,
                       stride, N);
  }
  return check_launch("sorted_foreground");
}

int rd_decode3d_bbox(const float* bbox_delta, const float* pc, float* out, int B, long N, int box_type, int is_bin,
                     void* stream) {
  RD_REQUIRE(bbox_delta && pc && out, RD_EINVAL, "decode3d: null pointer");
  RD_REQUIRE(B > 0 && N > 0, RD_ESHAPE, "decode3d: empty shape");
  RD_REQUIRE((is_bin && box_type == 7) || (!is_bin && box_type == 8), RD_ESHAPE,
             "decode3d: box_type %d with is_bin=%d (8 for regular, 7 for bin)", box_type, is_bin);
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)B * N;
  ProfScope ps(RD_PROF_DECODE, st);
  hipLaunchKernelGGL(decode3d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, bbox_delta, pc, out, n, box_type, is_bin);
  return check_launch("decode3d");
}

size_t rd_score_filter_workspace_bytes(long n) { return (size_t)((n + 255) / 256 + 8) * 4 + 256; }
int rd_score_filter_dets_batched(const float* scores, long scores_bstride, const float* boxes10, long boxes_bstride,
                                 long n, float min_score, float* dets, long dets_bstride, int* d_count, void* ws,
                                 size_t ws_bytes, int B, void* stream) {
  RD_REQUIRE(scores && boxes10 && dets && d_count && ws, RD_EINVAL, "score_filter: null pointer");
  RD_REQUIRE(n > 0 && B > 0, RD_ESHAPE, "score_filter: empty input");
  const size_t per = rd_score_filter_workspace_bytes(n);
  RD_REQUIRE(ws_bytes >= per * (size_t)B, RD_EWORKSPACE, "score_filter: workspace %zu < %zu", ws_bytes, per * (size_t)B);
  hipStream_t st = (hipStream_t)stream;
  int* blk = (int*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  const int nblk = (int)((n + 255) / 256);
  const long blk_bs = nblk + 8;                  // per-frame block-count arrays (ints apart)
  ProfScope ps(RD_PROF_WNMS, st);
  hipLaunchKernelGGL(filter_count_kernel, dim3(nblk, B), dim3(256), 0, st, scores, n, min_score, blk, scores_bstride, blk_bs);
  hipLaunchKernelGGL(filter_scan_kernel, dim3(B), dim3(256), 0, st, blk, nblk, d_count, blk_bs);
  hipLaunchKernelGGL(filter_scatter_kernel, dim3(nblk, B), dim3(256), 0, st, scores, boxes10, n, min_score, blk, dets,
                     scores_bstride, boxes_bstride, blk_bs, dets_bstride);
  return check_launch("score_filter");
}
int rd_score_filter_dets(const float* scores, const float* boxes10, long n, float min_score, float* dets, int* d_count,
                         void* ws, size_t ws_bytes, void* stream) {
  return rd_score_filter_dets_batched(scores, 0, boxes10, 0, n, min_score, dets, 0, d_count, ws, ws_bytes, 1, stream);
}

size_t rd_wnms_workspace_bytes(int Kcap) { return Kcap > 0 ? wnms_ws_bytes(Kcap) : 0; }
int rd_wnms_4c_batched(const float* dets, long dets_bstride, int Kcap, const int* d_count, const int* order,
                       long order_bstride, int tie_order, float thresh, float thresh_vote, int is3d, int hash_scale,
                       float* out_dets, long out_bstride, int* keep, long keep_bstride, int* d_nkeep, void* ws,
                       size_t ws_bytes, int B, void* stream) {
  RD_REQUIRE(dets && out_dets && keep && d_nkeep && ws, RD_EINVAL, "wnms_4c: null pointer");
  RD_REQUIRE(Kcap > 0 && Kcap <= RD_WNMS_MAX_K, RD_ESHAPE, "wnms_4c: Kcap %d not in [1, %d]", Kcap, RD_WNMS_MAX_K);
  RD_REQUIRE(B > 0, RD_ESHAPE, "wnms_4c: batch %d", B);
  // diagnostic bits of tie_order (include/rangedet_hip.h RD_WNMS_DIAG_*): per-call test aids, results do not depend on them
  const int diag = tie_order & ~0xff;
  tie_order &= 0xff;
  RD_REQUIRE(tie_order == RD_TIE_STABLE || tie_order == RD_TIE_REFERENCE, RD_EINVAL, "wnms_4c: tie_order %d", tie_order);
  RD_REQUIRE(!(diag & ~(RD_WNMS_DIAG_NO_SKIP | (0xff << 16) | (0x7f << 24))), RD_EINVAL, "wnms_4c: unknown diagnostic bits 0x%x", diag);
  const size_t per = rd_wnms_workspace_bytes(Kcap);
  RD_REQUIRE(ws_bytes >= per * (size_t)B, RD_EWORKSPACE, "wnms_4c: workspace %zu < %zu", ws_bytes, per * (size_t)B);
  RD_REQUIRE(order || B == 1 || tie_order == RD_TIE_REFERENCE, RD_EINVAL,
             "wnms_4c: the batched call needs an explicit processing order or RD_TIE_REFERENCE");
  hipStream_t st = (hipStream_t)stream;
  // (every kernel of the chain that may take more than 64 KB of dynamic LDS, once per device: no runtime-API call besides the launches
  //  themselves is left on the path, so the whole chain can be captured into a hipGraph -- pipeline.RangeDetPipeline(graph=True))
  static std::atomic<unsigned long long> seen_lds{0};
  once_per_device(seen_lds, [] {
    allow_big_lds(wnms_tie_order_kernel); allow_big_lds(wnms_scan_kernel); allow_big_lds(wnms_scan4_kernel); allow_big_lds(wnms_merge_kernel);
  });
  WnmsWs w = wnms_ws_carve(ws, Kcap);
  ProfScope ps(RD_PROF_WNMS, st);
  // consecutive frames use consecutive `per`-byte workspaces (per is a multiple of 256), so every carved array of
  // frame b sits b * per bytes after frame 0's
  WnmsBatch bs;
  bs.dets = dets_bstride; bs.order = order_bstride; bs.prep = (long)(per / 4); bs.words = (long)(per / 8);
  bs.ints = (long)(per / 4); bs.keep = keep_bstride; bs.out = out_bstride;
  const int* ord = order;
  if (!ord && tie_order == RD_TIE_REFERENCE) {   // the reference's own order: std::sort replayed on the device
    const size_t lds = TIE_STACK_BYTES + (Kcap <= TIE_LDS_K ? (size_t)Kcap * 8 + 3 * ((size_t)Kcap / 32 + 2) * 4 : 0);
    hipLaunchKernelGGL(wnms_tie_order_kernel, dim3(1, 1, B), dim3(256), lds, st, dets, Kcap, d_count, w.order, bs, (long)(per / 4),
                       w.scratch);
    ord = w.order;
    bs.order = (long)(per / 4);
  } else if (!ord) {  // device ordering: score descending, ties by row index ascending
    void* swa = (void*)(((uintptr_t)w.sort_ws + 255) & ~(uintptr_t)255);
    SortWs s = sort_ws_carve(swa, Kcap);
    hipLaunchKernelGGL(sort_keygen_dets_kernel, dim3((Kcap + 255) / 256), dim3(256), 0, st, dets, Kcap, d_count, s.keysA, s.idxA);
    int rc = radix_sort_pairs(s, Kcap, st);
    if (rc != RD_OK) return rc;
    if (hipMemcpyAsync(w.order, s.idxA, (size_t)Kcap * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return rd::fail(RD_EHIP, "wnms_4c: order copy");
    ord = w.order;
    bs.order = 0;
  }
  const int nb = (Kcap + 63) / 64;
  hipLaunchKernelGGL(wnms_prep_kernel, dim3((Kcap + 255) / 256, 1, B), dim3(256), 0, st, dets, ord, Kcap, d_count, w.prep, bs,
                     (float)hash_scale, w.novf);
  // RD_WNMS_DIAG_TILE_W(n) / RD_WNMS_DIAG_MERGE_LDS(n): per-call test aids that force the column-chunked scan and the merge overflow
  // path at small K (environment variables until round 5)
  const int e_tw = (diag >> 16) & 0xff, e_ml = (diag >> 24) & 0x7f;
  const int tile_w = std::min(w.nwcap, e_tw ? std::max(1, e_tw) : 256);
  const size_t scan_lds = ((size_t)w.nwcap + (size_t)64 * tile_w) * 8;
  // Two rounds.  The greedy scan only ever reads the thr / vote rows of boxes it KEEPS, and the highest-scoring boxes
  // suppress most of the rest: round 1 evaluates the pairs of the first R1 rows and scans them; round 2 evaluates pairs
  // only for the later rows that are still unsuppressed (compacted list) and resumes the scan.  Results are identical
  // to the one-round form -- the skipped rows are exactly the ones whose bits nobody reads.
  const int R1 = 256, nb1 = R1 / 64;
  const bool one_round = dev_switches().wnms_one_round;   // dev switch (tools/wnms_bench.py)
  const bool two = Kcap >= 4 * R1 && !one_round;
  // pair tiles are strided over a fixed number of single-wave workgroups per frame: one tile each at the pipeline's typical K
  // (1 - 2 k rows: <= 2048 tiles per round), grid-strided beyond that -- so the launch size does not grow with the capacity
  const int ct = dev_switches().wnms_ct == 32 ? 32 : dev_switches().wnms_ct == 16 ? 16 : 8;   // columns per pair tile
  const int pgrid = std::min(2048, std::max(64, nb * (64 / ct) * std::min(nb, 16)));
  // the rejection test of the pair kernel (k_wnms.h w_pair_skippable) is only sound for thresholds far above the noise the
  // reference's clipper returns on disjoint boxes (<= 4.2e-7 inside the test's domain, profiles/r04_nms_spurious_study.txt), and it
  // was characterised on the BEV value only: in 3-D mode the reference divides the clipped area x height overlap by a volume sum
  // that non-positive or cancelling heights can make arbitrarily small (nms.h:234-246), so there every pair is clipped.
  // RD_WNMS_DIAG_NO_SKIP (per call; dev builds also RD_WNMS_NO_SKIP): every pair clipped (A/B)
  const int allow_skip = thresh >= 1e-3f && thresh_vote >= 1e-3f && !is3d && !dev_switches().wnms_no_skip && !(diag & RD_WNMS_DIAG_NO_SKIP);
  auto pairs = [&](const int* rows, const int* nrows, const unsigned long long* supp, int rb_end) {
    auto k = dev_switches().wnms_bal ? (ct == 8 ? wnms_pairs_kernel<8, true> : ct == 16 ? wnms_pairs_kernel<16, true> : wnms_pairs_kernel<32, true>)
                                     : (ct == 8 ? wnms_pairs_kernel<8, false> : ct == 16 ? wnms_pairs_kernel<16, false> : wnms_pairs_kernel<32, false>);
    hipLaunchKernelGGL(k, dim3(pgrid, 1, B), dim3(64), 0, st, w.prep, Kcap, d_count, thresh, thresh_vote, is3d, w.thr, w.vote,
                       w.nwcap, bs, rows, nrows, supp, 0, rb_end, allow_skip);
  };
  pairs(nullptr, nullptr, nullptr, two ? nb1 : nb);
  // capacities up to 8 192 rows: the four-wave scan with grouped staging (RD_WNMS_SCAN1 / a forced tile width: the single-wave form)
  const bool scan4 = w.nwcap <= 128 && !e_tw && !dev_switches().wnms_scan1;
  const int tile_words = 64 * 128;
  const size_t scan4_lds = ((size_t)w.nwcap + tile_words) * 8;
  auto scan = [&](int c_begin, int c_end, unsigned long long* state) {
    if (scan4)
      hipLaunchKernelGGL(wnms_scan4_kernel, dim3(1, 1, B), dim3(256), scan4_lds, st, w.thr, w.snap, Kcap, d_count, w.nwcap, ord,
                         w.keep_q, keep, d_nkeep, bs, c_begin, c_end, state, tile_words);
    else
      hipLaunchKernelGGL(wnms_scan_kernel, dim3(1, 1, B), dim3(64), scan_lds, st, w.thr, w.snap, Kcap, d_count, w.nwcap, ord,
                         w.keep_q, keep, d_nkeep, bs, c_begin, c_end, state, tile_w);
  };
  scan(0, two ? nb1 : nb, two ? w.supp_state : (unsigned long long*)nullptr);
  if (two) {
    hipLaunchKernelGGL(wnms_alive_kernel, dim3(1, 1, B), dim3(256), 0, st, w.supp_state, Kcap, d_count, R1, w.alive, w.nalive, bs);
    pairs((const int*)w.alive, (const int*)w.nalive, (const unsigned long long*)w.supp_state, 0);
    scan(nb1, nb, w.supp_state);
  }
  // neighbourhood list of a kept row: 2 048 entries (16 KB) in LDS -- the lists are a few dozen entries long; a frame with more
  // candidates than that counts each row's list first and sends the rare longer one through the global-scratch kernel.  (With the
  // former 16 384 entries every single-wave workgroup asked for 65 KB: in the pipeline, where the conv workgroups hold 2 x 80 KB
  // per CU, the merge took 80 us instead of 21; +0.3 % frames/s)
  const int lds_cap = std::min(Kcap + 2, e_ml ? std::max(4, e_ml) : 2048);
  hipLaunchKernelGGL(wnms_merge_kernel, dim3(std::min(Kcap, 4096), 1, B), dim3(64), (size_t)lds_cap * 8, st, dets, ord, w.vote,
                     w.snap, Kcap, d_count, w.nwcap, w.keep_q, d_nkeep, out_dets, bs, lds_cap, w.ovf, w.novf);
  if (Kcap + 2 > lds_cap)
    hipLaunchKernelGGL(wnms_merge_big_kernel, dim3(1, 1, B), dim3(64), 0, st, dets, ord, w.vote, w.snap, Kcap, d_count, w.nwcap,
                       w.keep_q, out_dets, bs, (const int*)w.ovf, (const int*)w.novf, w.scratch);
  return check_launch("wnms_4c");
}
int rd_wnms_4c(const float* dets, int Kcap, const int* d_count, const int* order, int tie_order, float thresh,
               float thresh_vote, int is3d, int hash_scale, float* out_dets, int* keep, int* d_nkeep, void* ws,
               size_t ws_bytes, void* stream) {
  return rd_wnms_4c_batched(dets, 0, Kcap, d_count, order, 0, tie_order, thresh, thresh_vote, is3d, hash_scale, out_dets, 0,
                            keep, 0, d_nkeep, ws, ws_bytes, 1, stream);
}
int rd_single_overlap(const float* dets_a, const float* dets_b, long n, int is3d, float* out, void* stream) {
  RD_REQUIRE(n >= 0 && (n == 0 || (dets_a && dets_b && out)), RD_EINVAL, "single_overlap: bad arguments");
  if (n == 0) return RD_OK;
  hipLaunchKernelGGL(single_overlap_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, dets_a, dets_b, n,
                     is3d, out);
  return check_launch("single_overlap");
}
int rd_wnms_pair_skippable(const float* dets_a, const float* dets_b, long n, unsigned char* out, void* stream) {
  RD_REQUIRE(dets_a && dets_b && out && n >= 0, RD_EINVAL, "wnms_pair_skippable: null pointer");
  if (n == 0) return RD_OK;
  hipLaunchKernelGGL(pair_skippable_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, dets_a, dets_b, n, out);
  return check_launch("wnms_pair_skippable");
}
namespace rd {
__global__ void atan2f_kernel(const float* y, const float* x, long n, float* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fdlibm_atan2f(y[i], x[i]);
}
}  // namespace rd
int rd_edge_atan2f(const float* y, const float* x, long n, float* out, void* stream) {
  RD_REQUIRE(y && x && out && n >= 0, RD_EINVAL, "edge_atan2f: null pointer");
  if (n == 0) return RD_OK;
  hipLaunchKernelGGL(atan2f_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, x, n, out);
  return check_launch("edge_atan2f");
}
int rd_wnms_order_host(const float* dets_host, int K, int* order_host) {
  RD_REQUIRE(K >= 0 && (K == 0 || (dets_host && order_host)), RD_EINVAL, "wnms_order_host: bad arguments");
  // the reference's ordering, literally (nms.h:786-792): std::sort is unstable, so equal scores come out in
  // libstdc++ introsort order -- reproduced by running the same call on the host.
  std::iota(order_host, order_host + K, 0);
  std::sort(order_host, order_host + K, [&](int i, int j) { return dets_host[i * 12 + 11] > dets_host[j * 12 + 11]; });
  return RD_OK;
}

int rd_dets12_to_8_batched(const float* dets12, long dets12_bstride, int Mcap, const int* d_count, float* out8,
                           long out8_bstride, int B, void* stream) {
  RD_REQUIRE(dets12 && out8, RD_EINVAL, "dets12_to_8: null pointer");
  RD_REQUIRE(Mcap > 0 && B > 0, RD_ESHAPE, "dets12_to_8: empty input");
  hipLaunchKernelGGL(dets12_to_8_kernel, dim3((Mcap + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, dets12, Mcap, d_count,
                     out8, dets12_bstride, out8_bstride);
  return check_launch("dets12_to_8");
}
int rd_dets12_to_8(const float* dets12, int Mcap, const int* d_count, float* out8, void* stream) {
  return rd_dets12_to_8_batched(dets12, 0, Mcap, d_count, out8, 0, 1, stream);
}

int rd_gather_keep_scores(const float* score, long score_bstride, int k, const int* keep_idx, int max_keep, float* out, int B,
                          void* stream) {
  RD_REQUIRE(score && keep_idx && out, RD_EINVAL, "gather_keep_scores: null pointer");
  RD_REQUIRE(k > 0 && max_keep > 0 && B > 0, RD_ESHAPE, "gather_keep_scores: empty input");
  hipLaunchKernelGGL(gather_keep_scores_kernel, dim3((max_keep + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, score,
                     score_bstride, k, keep_idx, max_keep, out);
  return check_launch("gather_keep_scores");
}

int rd_rotated_iou_8pt(const float* boxes1, const float* boxes2, float* ious, long n1, long n2, void* stream) {
  RD_REQUIRE(boxes1 && boxes2 && ious, RD_EINVAL, "rotated_iou: null pointer");
  RD_REQUIRE(n1 > 0 && n2 > 0, RD_ESHAPE, "rotated_iou: empty input");
  hipLaunchKernelGGL(riou8_kernel, dim3((unsigned)((n1 * n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes1, boxes2, ious, n1, n2);
  return check_launch("rotated_iou_8pt");
}
int rd_batch_rotated_iou(const float* proposal, int p_stride, const float* gt_bbox, float* iou_map, int* argmax, int B, long N,
                         int n_gt, void* stream) {
  RD_REQUIRE(proposal && gt_bbox && iou_map, RD_EINVAL, "batch_rotated_iou: null pointer");
  RD_REQUIRE(B > 0 && B <= 65535 && N > 0 && n_gt > 0 && n_gt <= 256 && p_stride >= 8, RD_ESHAPE,
             "batch_rotated_iou: B %d, N %ld, n_gt %d (<= 256), proposal row %d floats (>= 8)", B, N, n_gt, p_stride);
  hipLaunchKernelGGL(batch_riou_kernel, dim3((unsigned)((N + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, proposal, p_stride,
                     N * (long)p_stride, gt_bbox, (long)n_gt * 8, iou_map, argmax, N, n_gt);
  return check_launch("batch_rotated_iou");
}
int rd_rotated_iou_7(const float* boxes1, const float* boxes2, float* ious, long n1, long n2, void* stream) {
  RD_REQUIRE(boxes1 && boxes2 && ious, RD_EINVAL, "rotated_iou_7: null pointer");
  if (n1 * n2 == 0) return RD_OK;
  hipLaunchKernelGGL(riou7_kernel, dim3((unsigned)((n1 * n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes1, boxes2, ious, n1, n2);
  return check_launch("rotated_iou_7");
}
int rd_batch_rotated_iou_3d(const float* proposal, int p_stride, const float* gt_bbox7, float* iou_map, int* argmax, int B, long N,
                            int n_gt, void* stream) {
  RD_REQUIRE(proposal && gt_bbox7 && iou_map, RD_EINVAL, "batch_rotated_iou_3d: null pointer");
  RD_REQUIRE(B > 0 && N > 0 && p_stride >= 10, RD_ESHAPE, "batch_rotated_iou_3d: proposal rows need the 8 corners + z0, z1 (stride %d)", p_stride);
  RD_REQUIRE(n_gt >= 1 && n_gt <= 256, RD_ESHAPE, "batch_rotated_iou_3d: n_gt %d (1..256; the config pads to 200)", n_gt);
  ProfScope ps(RD_PROF_DECODE, (hipStream_t)stream);
  hipLaunchKernelGGL(batch_riou3d_kernel, dim3((unsigned)((N + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, proposal, p_stride,
                     (long)N * p_stride, gt_bbox7, (long)n_gt * 7, iou_map, argmax, N, n_gt);
  return check_launch("batch_rotated_iou_3d");
}
int rd_batch_max_iou(const float* proposals, int p_stride, const float* gt8, float* out, long n, int n_gt, void* stream) {
  return rd_batch_rotated_iou(proposals, p_stride, gt8, out, nullptr, 1, n, n_gt, stream);
}

// ---- greedy 3-D NMS (_contrib_NMS3D) ----------------------------------------------------------------------------------
size_t rd_nms3d_workspace_bytes(long N, int B) {
  if (N <= 0 || N > (1 << 18) || B <= 0) return 0;
  return nms3d_frame_bytes((int)N) * (size_t)B;
}
int rd_nms3d(const float* boxes, int B, long N, float iou_thres, int max_keep, int normal_iou, int* keep_idx,
             float* bbox_after_nms, void* ws, size_t ws_bytes, void* stream) {
  RD_REQUIRE(boxes && keep_idx && bbox_after_nms && ws, RD_EINVAL, "nms3d: null pointer");
  RD_REQUIRE(B > 0 && B <= 65535 && N > 0 && N <= (1 << 18) && max_keep > 0, RD_ESHAPE, "nms3d: B %d, N %ld (<= 262144), max_keep %d", B, N, max_keep);
  RD_REQUIRE(ws_bytes >= rd_nms3d_workspace_bytes(N, B), RD_EWORKSPACE, "nms3d: workspace %zu < %zu bytes", ws_bytes, rd_nms3d_workspace_bytes(N, B));
  RD_REQUIRE(((uintptr_t)ws & 7) == 0, RD_EINVAL, "nms3d: workspace must be 8-byte aligned");
  Nms3dArgs a;
  a.boxes = boxes; a.ws = (unsigned char*)ws; a.ws_frame = nms3d_frame_bytes((int)N);
  a.keep = keep_idx; a.out = bbox_after_nms;
  a.N = (int)N; a.ncw = (int)((N + 63) / 64); a.max_keep = max_keep; a.normal_iou = normal_iou ? 1 : 0; a.thresh = iou_thres;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(RD_PROF_WNMS, st);
  const int gi = (std::max(std::max(a.ncw, max_keep), 2) + 255) / 256;
  hipLaunchKernelGGL(nms3d_init_kernel, dim3(gi, B), dim3(256), 0, st, a);
  for (int r0 = 0, blk = 64; r0 < a.N; r0 += blk, blk = std::min(2 * blk, NMS3D_RB)) {
    const int rows = std::min(blk, a.N - r0);
    hipLaunchKernelGGL(nms3d_pairs_kernel, dim3((a.N + NMS3D_SEG - 1) / NMS3D_SEG, rows, B), dim3(256), 0, st, a, r0);
    hipLaunchKernelGGL(nms3d_scan_kernel, dim3(B), dim3(256), (size_t)a.ncw * 8, st, a, r0, blk);
  }
  return check_launch("nms3d");
}

// ---- target assignment (processing_cxx.assign3D_v2 / get_point_num) --------------------------------------------------
int rd_assign3d_v2(const float* pc, const float* bbox, const float* bbox_center, const float* bbox_radius, const float* mask,
                   const float* is_in_nlz, long N, int M, float max_x, float min_x, float max_y, float min_y, float max_z,
                   float min_z, float max_dist, int* out, void* stream) {
  RD_REQUIRE(pc && bbox && bbox_center && bbox_radius && mask && is_in_nlz && out, RD_EINVAL, "assign3d_v2: null pointer");
  RD_REQUIRE(N > 0 && M > 0 && M <= ASSIGN_MAX_BOXES, RD_ESHAPE, "assign3d_v2: N %ld, M %d (1..%d)", N, M, ASSIGN_MAX_BOXES);
  AssignArgs a;
  a.pc = pc; a.bbox = bbox; a.center = bbox_center; a.radius = bbox_radius; a.mask = mask; a.nlz = is_in_nlz; a.out = out;
  a.N = N; a.M = M;
  a.max_x = max_x; a.min_x = min_x; a.max_y = max_y; a.min_y = min_y; a.max_z = max_z; a.min_z = min_z; a.max_dist = max_dist;
  hipLaunchKernelGGL(assign3d_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), (size_t)M * 14 * 4, (hipStream_t)stream, a);
  return check_launch("assign3d_v2");
}
size_t rd_get_point_num_workspace_bytes(void) { return POINT_NUM_MAX_BOXES * sizeof(int); }
int rd_get_point_num(const float* bbox_inds, long N, float* out, void* ws, size_t ws_bytes, void* stream) {
  RD_REQUIRE(bbox_inds && out && ws, RD_EINVAL, "get_point_num: null pointer");
  RD_REQUIRE(N > 0, RD_ESHAPE, "get_point_num: N %ld", N);
  RD_REQUIRE(ws_bytes >= rd_get_point_num_workspace_bytes(), RD_EWORKSPACE, "get_point_num: workspace %zu < %zu bytes", ws_bytes,
             rd_get_point_num_workspace_bytes());
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(ws, 0, rd_get_point_num_workspace_bytes(), st) != hipSuccess) return fail(RD_EHIP, "get_point_num: memset failed");
  const unsigned g = (unsigned)std::min<long>((N + 255) / 256, 1024);
  hipLaunchKernelGGL(point_num_count_kernel, dim3(g), dim3(256), 0, st, bbox_inds, N, (int*)ws);
  hipLaunchKernelGGL(point_num_gather_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, bbox_inds, N, (const int*)ws, out);
  return check_launch("get_point_num");
}

// ---- input transform chain ----------------------------------------------------------------------------------------
int rd_input_transform(const float* range_image, const float* pc_vehicle_frame, const float* inclination,
                       const rd_input_norm_t* norm_host, int B, int H, int W, int Hp, int Wp, float* input_data,
                       float* coord_s1, float* pc_s1, float* pc_s2, float* pc_s4, float* mask_s1, float* mask_s2,
                       float* mask_s4, void* stream) {
  RD_REQUIRE(range_image && pc_vehicle_frame && inclination && norm_host && input_data && coord_s1 && pc_s1 && pc_s2 && pc_s4 &&
                 mask_s1 && mask_s2 && mask_s4, RD_EINVAL, "input_transform: null pointer");
  RD_REQUIRE(B > 0 && H > 2 && W > 2 && Hp >= H && Wp >= W && Wp % 4 == 0, RD_ESHAPE,
             "input_transform: shape (%d,%d,%d) -> (%d,%d)", B, H, W, Hp, Wp);
  InputArgs a;
  a.ri = range_image; a.pc = pc_vehicle_frame; a.incl = inclination;
  a.data = input_data; a.coord = coord_s1;
  a.pcs[0] = pc_s1; a.pcs[1] = pc_s2; a.pcs[2] = pc_s4;
  a.msk[0] = mask_s1; a.msk[1] = mask_s2; a.msk[2] = mask_s4;
  a.B = B; a.H = H; a.W = W; a.Hp = Hp; a.Wp = Wp;
  a.n = *norm_host;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(RD_PROF_LAYOUT, st);
  hipLaunchKernelGGL(input_transform_kernel, dim3((unsigned)(((long)Hp * Wp + 255) / 256), B), dim3(256), 0, st, a);
  return check_launch("input_transform");
}

// ---- profiling -------------------------------------------------------------------------------------------------
// dev only (not part of include/rangedet_hip.h): the phase trace of the conv kernels.  rd_dev_conv_trace_set registers a
// caller-owned, zeroed device buffer of at least 2^20 64-bit words (nullptr: tracing off); rd_dev_conv_trace_read copies the first n
// words of it to the host.
int rd_dev_conv_trace_set(unsigned long long* device_buf, long nwords) {
  if (device_buf && nwords < (long)CONV_TRACE_CAP) return RD_EINVAL;
  conv_trace_slot() = device_buf;
  return RD_OK;
}
int rd_dev_conv_trace_read(unsigned long long* out, long n) {
  if (!conv_trace_buf() || n > (long)CONV_TRACE_CAP) return RD_EINVAL;
  return hipMemcpy(out, conv_trace_buf(), n * 8, hipMemcpyDeviceToHost) == hipSuccess ? RD_OK : RD_EHIP;
}
int rd_prof_enable(int on) {
  Prof& p = prof();
  std::lock_guard<std::mutex> g(p.mu);
  p.on = on != 0;
  return RD_OK;
}
int rd_prof_reset(void) {
  Prof& p = prof();
  std::lock_guard<std::mutex> g(p.mu);
  p.drain();
  for (int k = 0; k < RD_PROF_NKINDS; ++k) { p.total[k] = 0; p.count[k] = 0; }
  return RD_OK;
}
int rd_prof_get(int kind, double* total_ms, long* launches) {
  RD_REQUIRE(kind >= 0 && kind < RD_PROF_NKINDS && total_ms && launches, RD_EINVAL, "prof_get: bad arguments");
  Prof& p = prof();
  std::lock_guard<std::mutex> g(p.mu);
  p.drain();
  *total_ms = p.total[kind];
  *launches = p.count[kind];
  return RD_OK;
}

}  // extern "C"
