/* rangedet_hip.h -- C ABI of librangedet_hip.so: the MI355X (gfx950) RangeDet inference hot path.
 *
 * Every entry point replaces one piece of the reference's native/operator surface for this path
 * (citations relative to /root/reference; see INTEGRATION.md for the binding a maintainer would add):
 *
 *   rd_meta_kernel_fwd        MetaKernel.meta_baseline_bias + BN/ReLU/1x1/BN/ReLU
 *                             rangedet/symbol/backbone/meta_kernel.py:166-240, dla_backbone.py:92-97
 *   rd_conv2d_bn_act          mx.sym.Convolution + BatchNorm (+ReLU, +residual)   mxnext/simple.py:123-158,
 *                             mxnext/complicate.py:26-45, dla_backbone.py:18-56, head/builder.py:221-240
 *   rd_conv3x3_bn_act_ex      BasicBlock conv2 (+ stride (1,2), + projection shortcut)   dla_backbone.py:18-56,139-143
 *   rd_block64_bn_act         a whole 64-channel BasicBlock (conv1 + conv2 + shortcut) in one launch   dla_backbone.py:18-56
 *   rd_conv3x3_bn_act_pair,   the cls and the reg tower conv i of a head level in ONE launch (the last pair with the towers'
 *   rd_conv2d_bn_act_head_out_pair   1x1 output convs)                            head/builder.py:221-261
 *   rd_deconv2d_bn_act,       mx.sym.Deconvolution + BatchNorm + ReLU + add       mxnext/simple.py:545-580,
 *   rd_deconv2d_bn_act_all,   (one call per output phase / all phases in one launch / phase pairs as 128-channel
 *   rd_deconv2d_bn_act_pairs  problems)                                           dla_backbone.py:117-127
 *   rd_head_out               1x1 logit / delta convs + cast + per-class flatten  head/builder.py:242-261,99-154
 *   rd_sorted_foreground      Custom op 'get_sorted_foreground'                   operator_py/get_sorted_foreground.py:11-40
 *   rd_decode3d_bbox          _contrib_Decode3DBbox                               operator_cxx/contrib/decode_3d_bbox-inl.h:169-305
 *   rd_score_filter_dets      score filter + bbox3d_10dim_to_11dim                tools/test.py:56-81,200-209
 *   rd_wnms_4c                processing_cxx.wnms_4c                              operator_cxx/src_cxx/nms.h:452-577,781-794
 *   rd_single_overlap         OverlapChecker::single_overlap                      operator_cxx/src_cxx/nms.h:195-249
 *   rd_wnms_order_host        the std::sort ordering of point4_wnms_4c            operator_cxx/src_cxx/nms.h:786-792
 *   rd_dets12_to_8            bbox3d_12dim_to_8dim                                tools/test.py:43-53
 *   rd_rotated_iou_8pt        _contrib_RotatedIOU (8-point boxes)                 operator_cxx/contrib/rotated_iou-inl.h:509-547
 *   rd_batch_max_iou          Custom op 'batch_rotated_iou' ('bev')               operator_py/batch_rotated_iou.py:11-49
 *   rd_nms3d                  _contrib_NMS3D (the wnms=False branch)              operator_cxx/contrib/nms_3d.cu:380-534,
 *                             head/builder.py:530-534
 *   rd_assign3d_v2            processing_cxx.assign3D_v2                          operator_cxx/src_cxx/assigner.h:11-85
 *   rd_get_point_num          processing_cxx.get_point_num                        operator_cxx/src_cxx/assigner.h:87-109
 *   rd_input_transform        test-time input transform chain                     rangedet/core/input.py:14-42,89-229,522-624
 *
 * Conventions
 *   - every function returns an int status (RD_OK == 0, negative = error); nothing aborts or exits.
 *     rd_last_error_string() gives a thread-local message for the last failure.
 *   - all pointers are DEVICE pointers unless the name ends in _host; the caller owns every buffer,
 *     including workspaces (size them with the *_workspace_bytes functions).  The library never allocates
 *     device memory and never keeps a pointer past the call.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); calls are re-entrant.
 *   - activations inside the library are channels-last: [B][H][W][Cstride] with the used channels at
 *     [coff, coff+C); element type RD_F32 (float), RD_BF16 (raw uint16 bfloat16) or RD_F16 (raw uint16 IEEE half).
 *     rd_nchw_to_nhwc / rd_nhwc_to_nchw convert at the reference's NCHW float32 boundary.
 */
#ifndef RANGEDET_HIP_H_
#define RANGEDET_HIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RD_OK 0
#define RD_EINVAL (-1)     /* bad argument value                       */
#define RD_ESHAPE (-2)     /* shape the kernels do not support / mismatch (mirrors the reference's CHECKs) */
#define RD_EWORKSPACE (-3) /* workspace too small                      */
#define RD_EHIP (-4)       /* HIP runtime error                        */

#define RD_F32 0
#define RD_BF16 1
#define RD_F16 2  /* IEEE binary16: the reference's own mixed-precision type (config fp16 = True, dla_backbone.py:136-137 to_fp16,
                   * builder.py:257-261 to_fp32); every 16-bit layout is shared with RD_BF16 */

/* epilogue flags of the conv family */
#define RD_RELU_PRE 1  /* ReLU directly after the BN affine (before the residual add)        */
#define RD_ADD 2       /* add `residual`                                                     */
#define RD_RELU_POST 4 /* ReLU after the residual add                                        */
#define RD_SCALE_FOLDED 8 /* 16-bit 3x3 family only: the packer folded the BatchNorm scale into the weights (fold_scale argument of
                           * the packers); `scale` must be NULL.  The shift then enters the accumulators through one extra MFMA
                           * per accumulator and the epilogue has no multiply-add (what the production lowering uses) */
#define RD_MFMA16 16 /* rd_conv3x3_bn_act_ex (with RD_SCALE_FOLDED; stride 1, cout 128, cin a multiple of 32, no fused shortcut): w_packed is
                      * the image of rd_pack_conv3x3_m16_host and the launch issues v_mfma_f32_16x16x32 instead of 32x32x16 -- the same
                      * sums in the same per-channel order of the 32-channel chunks and taps; under the power cap of the part the
                      * matrix cores sustain more of them.  Like RD_SCALE_FOLDED a property of the packed image
                      * that the caller passes along: the library keeps no state per image */

int rd_version(void);
const char* rd_last_error_string(void);

/* ---- layout -------------------------------------------------------------------------------------- */
/* src NCHW float32 (B,C,H,W) -> dst channels-last [B][H][W][dst_cstride] at channel offset dst_coff.
 * Channels [C, C+zero_pad) are written as zeros (zero_pad may be 0). */
int rd_nchw_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int dst_cstride, int dst_coff,
                    int zero_pad, int dst_dtype, void* stream);
int rd_nhwc_to_nchw(const void* src, float* dst, int B, int C, int H, int W, int src_cstride, int src_coff,
                    int src_dtype, void* stream);

/* `rows` rows of `bytes` bytes: row r from src + r*src_row_bytes to dst + r*dst_row_bytes + dst_offset_bytes (device to
 * device, on `stream`).  The per-level point / mask variables are concatenated with it (builder.py:454-458: concat of
 * pc_vehicle_frame_s{1,2,4} / range_image_mask_s{1,2,4} along the point axis), one call per level for the whole batch. */
int rd_copy_rows(const void* src, long src_row_bytes, void* dst, long dst_row_bytes, long dst_offset_bytes, long bytes,
                 int rows, void* stream);

/* ---- weight packing (HOST, no GPU needed) ---------------------------------------------------------- */
/* Packed conv weights: [ntaps][nchunk][Cout][8 slots * (16/elem) channels], zero padded; one k-chunk =
 * 8 slots of 16 bytes.  rd_conv_packed_bytes gives the size.  w_oihw_host is (Cout,Cin,KH,KW) float32
 * (mx Convolution layout); deconv weight is (Cin,Cout,KH,KW) (mx Deconvolution layout) and is packed per
 * output phase (phase = ow % stride_w) -- rd_deconv_phase_taps tells how many taps a phase's PACKED image has: the
 * phase's own tap count (<= 9, in ascending (dh, dw) order; more is RD_ESHAPE), except that a phase whose taps lie
 * inside the 3x3 window without being three rows x two adjacent columns is packed as all 9 window taps (zeros for the
 * absent ones). */
/* (every packed image ends in 256 zero bytes, included in the size: the persistent 3x3 kernel reads its padding pixels from
 * them, so the library holds no device memory of its own; upload the image whole) */
size_t rd_conv_packed_bytes(int ntaps, int cin, int cout, int dtype);
int rd_pack_conv_weight_host(const float* w_oihw_host, int cout, int cin, int kh, int kw, int dtype,
                             void* packed_host);
int rd_deconv_phase_taps(int kh, int kw, int stride_w, int pad_w, int phase);
int rd_pack_deconv_weight_host(const float* w_iohw_host, int cin, int cout, int kh, int kw, int stride_w,
                               int pad_w, int phase, int dtype, void* packed_host);
/* the same with fold_scale_host (cout) multiplied into the weights before rounding (for RD_SCALE_FOLDED launches; NULL = none) */
int rd_pack_deconv_weight_folded_host(const float* w_iohw_host, const float* fold_scale_host, int cin, int cout, int kh, int kw,
                                      int stride_w, int pad_w, int phase, int dtype, void* packed_host);

/* ---- conv family ------------------------------------------------------------------------------------ */
/* y = act( scale[c]*conv(x, w) + shift[c] (+ residual) ).  stride_h == 1 always (the backbone strides W
 * only, dla_backbone.py:139-143); kernel (kh,kw) in {(1,1),(3,3)}; pad = (k-1)/2 as mxnext/simple.py:131-135.
 * cout in {64,128}.  x: [B][H][Win][x_cstride] channels [x_coff, x_coff+cin); cin is rounded up internally to
 * the packing granule and the buffer must hold zeros there. */
int rd_conv2d_bn_act(const void* x, int x_cstride, int x_coff, const void* w_packed, const float* scale,
                     const float* shift, const void* residual, int r_cstride, int r_coff, void* y,
                     int y_cstride, int y_coff, int B, int H, int Win, int cin, int cout, int kh, int kw,
                     int stride_w, int flags, int dtype, void* stream);
/* Second conv of a BasicBlock in its extended forms (dla_backbone.py:18-56; dtype RD_BF16 or RD_F16, persistent kernel):
 *   stride_w 2  (`conv2` of the first unit of a down-sampling stage, dla_backbone.py:139-143): computed on the pixel-pair view
 *               of x -- [H][Win][cs] with Win even read as [H][Win/2][2*cs] -- as a stride-1 conv with six taps, so only the
 *               stored pixels are computed; weights packed by rd_pack_conv3x3_ex_host(stride_w = 2, x_cstride);
 *   sc_x != NULL  the block's projection shortcut  BN(Convolution 1x1 (stride_w) of the block input)  (dla_backbone.py:44-51)
 *               is accumulated onto the conv's accumulators in the epilogue (no shortcut tensor, no second launch).  The two
 *               BatchNorm scales must be FOLDED into the two weight sets (fold_scale of the packers), `scale` = NULL and
 *               `shift` = shift_conv + shift_shortcut; RD_ADD is implied, `residual` must be NULL.
 * Otherwise the contract of rd_conv2d_bn_act (3x3, pad 1).  sc_x: [B][H][Win][sc_cstride], channels [sc_coff, sc_coff+sc_cin). */
size_t rd_conv3x3_ex_packed_bytes(int cin, int cout, int stride_w, int x_cstride);
int rd_pack_conv3x3_ex_host(const float* w_oihw_host, const float* fold_scale_host, int cout, int cin, int stride_w,
                            int x_cstride, int dtype, void* packed_host);
/* weights of an RD_MFMA16 launch (same size as rd_conv3x3_ex_packed_bytes(cin, 128, 1, cin)); rd_conv3x3_mfma16_ok: 1 if the library has that
 * launch form for a stride_w conv cin -> cout at width W (fused_output_conv: through rd_conv2d_bn_act_head_out) -- the caller asks before
 * it packs; a launch with RD_MFMA16 where the answer is 0 fails with RD_ESHAPE */
int rd_pack_conv3x3_m16_host(const float* w_oihw_host, const float* fold_scale_host, int cout, int cin, int dtype, void* packed_host);
int rd_conv3x3_mfma16_ok(int cin, int cout, int stride_w, int W, int fused_output_conv);
size_t rd_conv1x1_sc_packed_bytes(int cin, int cout);
int rd_pack_conv1x1_sc_host(const float* w_oi_host, const float* fold_scale_host, int cout, int cin, int dtype,
                            void* packed_host);
int rd_conv3x3_bn_act_ex(const void* x, int x_cstride, int x_coff, const void* w_packed, const float* scale, const float* shift,
                         const void* residual, int r_cstride, int r_coff, const void* sc_x, int sc_cstride, int sc_coff,
                         int sc_cin, const void* sc_w_packed, void* y, int y_cstride, int y_coff, int B, int H, int Win, int cin,
                         int cout, int stride_w, int flags, int dtype, void* stream);
/* A whole 64-channel BasicBlock (rangedet/symbol/backbone/dla_backbone.py:18-56: conv1 3x3 + BN + ReLU, conv2 3x3 + BN, + shortcut, ReLU;
 * stride 1, cin -> 64 -> 64) as ONE launch -- replaces the two rd_conv3x3_bn_act_ex calls of the block; the intermediate tensor is never
 * written to HBM (it lives in LDS as conv2's halo image) and the results are BIT-IDENTICAL to the two calls.
 *   cin == 64, sc_w_packed == NULL  identity shortcut (units 2.. of a stage):  y = relu(BN2(conv2(relu(BN1(conv1(x))))) + x)
 *   cin == 64, sc_w_packed != NULL  projection shortcut (unit 1, dla_backbone.py:44-51): + BNs(conv1x1(x)) instead of + x; sc_w_packed =
 *                        rd_pack_conv1x1_sc_host(cin -> 64, fold_scale = the shortcut BatchNorm's scale), shift2 = conv2's + the shortcut's.
 *   cin <= 16            the network's first block (res1_unit1: the 8-channel range image, or 5 channels for KITTI): conv1 runs as five
 *                        two-tap steps on one 16-channel k-slot; the projection shortcut is required.
 * w_packed = rd_pack_block64_host (HOST): the 3x3 weights (64, cin, 3, 3) and (64, 64, 3, 3) with their BatchNorm scales folded in;
 * shift1 / shift2 (64) floats on the device.  x: [B][H][W][x_cstride] channels [x_coff, x_coff + cin) -- the channels up to the next
 * multiple of 16 must exist in the buffer and be zero --, y: channels [y_coff, y_coff + 64); 16-bit types only. */
size_t rd_block64_packed_bytes(int cin);
int rd_pack_block64_host(const float* w1_oihw_host, const float* fold_scale1_host, const float* w2_oihw_host,
                         const float* fold_scale2_host, int cin, int dtype, void* packed_host);
int rd_block64_bn_act(const void* x, int x_cstride, int x_coff, int cin, const void* w_packed, const float* shift1, const float* shift2,
                      const void* sc_w_packed, void* y, int y_cstride, int y_coff, int B, int H, int W, int dtype, void* stream);
/* The same block (64 input channels) issuing v_mfma_f32_16x16x32 instead of 32x32x16 (round 6; see RD_MFMA16): w_packed from
 * rd_pack_block64_m16_host (rd_block64_packed_bytes(64) bytes), sc_w_packed (projection shortcut 64 -> 64, or NULL) from
 * rd_pack_conv1x1_sc_m16_host (rd_conv1x1_sc_packed_bytes(64, 64) bytes); results identical to rd_block64_bn_act */
int rd_pack_block64_m16_host(const float* w1_oihw_host, const float* fold_scale1_host, const float* w2_oihw_host,
                             const float* fold_scale2_host, int dtype, void* packed_host);
int rd_pack_conv1x1_sc_m16_host(const float* w_oi_host, const float* fold_scale_host, int dtype, void* packed_host);
int rd_block64_m16_bn_act(const void* x, int x_cstride, int x_coff, const void* w_packed, const float* shift1, const float* shift2,
                          const void* sc_w_packed, void* y, int y_cstride, int y_coff, int B, int H, int W, int dtype, void* stream);

/* Last conv of a head tower (3x3, cout 128, BN + ReLU, RD_BF16 or RD_F16) FUSED with the tower's 1x1 output conv (head/builder.py:221-261:
 * rpn_{cls,reg}_conv_3 + BN + ReLU, then rpn_cls_logit / rpn_reg_delta with bias): the 128-channel result is consumed in
 * the epilogue and never written.  out[b*out_batch_stride + (n_off + h*W + w)*nout + o], float32, like rd_head_out; nout <= 8.
 * head_w_packed: rd_pack_head_weight_host(w (nout, 128) row-major float32) -> rd_head_packed_bytes() bytes (bf16 hi + lo
 * pairs, so the fp32 weights keep their precision).  Same numbers as rd_conv2d_bn_act followed by rd_head_out up to fp32
 * summation order. */
size_t rd_head_packed_bytes(void);
int rd_pack_head_weight_host(const float* w, int nout, int cin, int dtype, void* out_host);
/* ... of an RD_MFMA16 launch of rd_conv2d_bn_act_head_out (flags carry RD_MFMA16, w_packed from rd_pack_conv3x3_m16_host): head_w_packed must be
 * THIS image (rd_head_m16_packed_bytes() = 8 KB: 16-row fragments in the order the conv's epilogue holds the channels) -- the output conv
 * then reads the rounded tower activations straight from the registers */
size_t rd_head_m16_packed_bytes(void);
int rd_pack_head_weight_m16_host(const float* w, int nout, int cin, int dtype, void* out_host);
int rd_conv2d_bn_act_head_out(const void* x, int x_cstride, int x_coff, const void* w_packed, const float* scale,
                              const float* shift, int B, int H, int W, int cin, int flags, const void* head_w_packed,
                              const float* head_bias, float* out, long out_batch_stride, long n_off, int nout, int dtype,
                              void* stream);

/* The cls and the reg tower of a head level (head/builder.py:221-240: rpn_cls_conv_i / rpn_reg_conv_i, i = 0..3, the same
 * 3x3 conv + BN + ReLU shape twice on every level) as ONE launch: two problems -- input, packed weights (rd_pack_conv3x3_ex_host
 * with fold_scale), shift, output each -- that share B, H, W, cin, cout = 128, the channel strides and flags (RD_SCALE_FOLDED
 * required, no residual).  Same numbers as two rd_conv3x3_bn_act_ex / rd_conv2d_bn_act_head_out calls; the launch has twice
 * the tiles per resident workgroup (half the tail round) and one pipeline fill / drain instead of two.
 * rd_conv2d_bn_act_head_out_pair: the towers' LAST convs, each fused with its own 1x1 output conv (rpn_cls_logit: nout0,
 * rpn_reg_delta: nout1; head_w*_packed from rd_pack_head_weight_host; outputs as in rd_conv2d_bn_act_head_out, same n_off). */
int rd_conv3x3_bn_act_pair(const void* x0, int x0_coff, const void* w0_packed, const float* shift0, void* y0, int y0_coff,
                           const void* x1, int x1_coff, const void* w1_packed, const float* shift1, void* y1, int y1_coff,
                           int x_cstride, int y_cstride, int B, int H, int W, int cin, int flags, int dtype, void* stream);
int rd_conv2d_bn_act_head_out_pair(const void* x0, int x0_coff, const void* w0_packed, const float* shift0, const void* head_w0_packed,
                                   const float* head_bias0, float* out0, long out0_batch_stride, int nout0,
                                   const void* x1, int x1_coff, const void* w1_packed, const float* shift1, const void* head_w1_packed,
                                   const float* head_bias1, float* out1, long out1_batch_stride, int nout1,
                                   int x_cstride, long n_off, int B, int H, int W, int cin, int flags, int dtype, void* stream);

/* 3x3 stride-1 conv + BatchNorm (+ReLU) over the channel concatenation [x1 (cin1, a multiple of 32) | x2 (cin2, a multiple of 8)]
 * of two tensors of the same B x H x W -- the concat itself (mx.sym.concat of the range image with the agg3 feature map,
 * dla_backbone.py:153-154, consumed by the level-0 tower convs head/builder.py:221-240) is never written: each tensor keeps its own
 * channel stride.  16-bit types, RD_SCALE_FOLDED weights: w_packed = rd_pack_conv3x3_cat_host (HOST) of the (cout, cin1 + cin2, 3, 3)
 * weight in that channel order, rd_conv3x3_cat_packed_bytes long (with cin1 = 64, cin2 <= 16 the x2 chunk runs as five two-tap
 * steps on its one 16-channel slot instead of nine steps of two slots).  No residual. */
size_t rd_conv3x3_cat_packed_bytes(int cin1, int cin2, int cout);
int rd_pack_conv3x3_cat_host(const float* w, const float* fold_scale, int cout, int cin1, int cin2, int dtype, void* out_host);
int rd_conv3x3_bn_act_cat(const void* x1, int x1_cstride, int x1_coff, int cin1, const void* x2, int x2_cstride, int x2_coff, int cin2,
                          const void* w_packed, const float* shift, void* y, int y_cstride, int y_coff, int B, int H, int W, int cout,
                          int flags, int dtype, void* stream);

/* Transposed conv, kernel (3,kw), stride (1,stride_w), pad (1,pad_w); one call per output phase. */
int rd_deconv2d_bn_act(const void* x, int x_cstride, int x_coff, const void* w_packed_phase,
                       const float* scale, const float* shift, const void* residual, int r_cstride,
                       int r_coff, void* y, int y_cstride, int y_coff, int B, int H, int Win, int cin,
                       int cout, int kh, int kw, int stride_w, int pad_w, int phase, int flags, int dtype,
                       void* stream);

/* The same transposed conv, ALL stride_w output phases in ONE launch (16-bit types, RD_SCALE_FOLDED weights): the tile list is
 * (tile, phase) with a tile's phases computed back to back by one workgroup, so the input is read from HBM once instead of once
 * per phase launch (mx.sym.Deconvolution + BatchNorm + ReLU + add, mxnext/simple.py:545-580, dla_backbone.py:117-127).
 * w_packed_all: the stride_w images of rd_pack_deconv_weight_folded_host, phase p at byte offset p * w_phase_bytes.
 * rd_deconv2d_all_phases_ok: 1 if this (kernel, stride, pad, cout, dtype) can run in that form -- every phase a 3 x 2 tap set
 * inside the 3 x 3 window and Wout == stride_w * Win, which k(3,8) s4 p2 and k(3,4) s2 p1 are -- else 0 (call per phase). */
int rd_deconv2d_all_phases_ok(int kh, int kw, int stride_w, int pad_w, int cout, int dtype);
int rd_deconv2d_bn_act_all(const void* x, int x_cstride, int x_coff, const void* w_packed_all, long w_phase_bytes,
                           const float* shift, const void* residual, int r_cstride, int r_coff, void* y, int y_cstride,
                           int y_coff, int B, int H, int Win, int cin, int cout, int kh, int kw, int stride_w, int pad_w,
                           int flags, int dtype, void* stream);

/* The same launch with output phases 2p and 2p+1 computed as ONE 128-channel problem (cout 64, even stride, both phases of a pair
 * on the same two input columns -- k(3,8) s4 p2: dla_backbone.py:117-127 'agg1'): in the output seen as [H][Win][stride_w * cout] a
 * pair's channels are neighbours, so it is a 3 x 2-tap conv with 128 outputs and runs the cout-128 form of the kernel (bit-identical
 * to rd_deconv2d_bn_act_all).  y / residual: dense cout-channel tensors (cstride == cout, coff 0).
 * w_packed_pairs: stride_w / 2 images of rd_pack_deconv_phase_pair_host (phase images 2p, 2p+1 of rd_pack_deconv_weight_folded_host
 * interleaved; 2 x the bytes of one phase image each), pair p at byte offset p * w_pair_bytes.  shift2: the layer's shift twice
 * (2 * cout values).  rd_deconv2d_phase_pairs_ok: 1 if (kernel, stride, pad, cout, dtype) has this form, else 0. */
int rd_deconv2d_phase_pairs_ok(int kh, int kw, int stride_w, int pad_w, int cout, int dtype);
int rd_pack_deconv_phase_pair_host(const void* phase_a, const void* phase_b, int cin, int dtype, void* out);
int rd_deconv2d_bn_act_pairs(const void* x, int x_cstride, int x_coff, const void* w_packed_pairs, long w_pair_bytes,
                             const float* shift2, const void* residual, int r_cstride, int r_coff, void* y, int y_cstride,
                             int y_coff, int B, int H, int Win, int cin, int cout, int kh, int kw, int stride_w, int pad_w,
                             int flags, int dtype, void* stream);

/* 1x1 conv with bias to nout <= 8 float32 outputs per pixel, written flattened: out[(n_off + h*W + w)*nout + o]
 * (== the (B, N, nout) tensor after sep_level_type's reshape/transpose/concat; nout == 1 gives (B, N)).
 * w: (nout, cin) float32 device, bias: (nout).  out_batch_stride in elements. */
int rd_head_out(const void* x, int x_cstride, int x_coff, const float* w, const float* bias, float* out,
                long out_batch_stride, long n_off, int B, int H, int W, int cin, int nout, int dtype,
                void* stream);

/* ---- Meta-Kernel unit ------------------------------------------------------------------------------- */
/* Packed parameter block for rd_meta_kernel_fwd (HOST).  Inputs in the reference's layouts:
 * w0 (32,3) b0 (32) w1 (64,32) b1 (64) [mlp0/mlp1 1x1 convs], s1,t1 (576) [BN folded, index c*9+k],
 * agg (64,576) [aggregation_conv1], s2,t2 (64).  */
size_t rd_meta_packed_bytes(int dtype);
int rd_pack_meta_host(const float* w0, const float* b0, const float* w1, const float* b1, const float* s1,
                      const float* t1, const float* agg, const float* s2, const float* t2, int dtype,
                      void* packed_host);
/* data: [B][H][W][d_cstride] (64 channels at d_coff), coord: NCHW float32 (B,3,H,W) as the reference's
 * coord_s1 variable, y: [B][H][W][y_cstride] 64 channels at y_coff. */
int rd_meta_kernel_fwd(const void* data, int d_cstride, int d_coff, const float* coord_nchw,
                       const void* packed, void* y, int y_cstride, int y_coff, int B, int H, int W, int dtype,
                       void* stream);

/* ---- post-processing -------------------------------------------------------------------------------- */
/* bytes for ONE batch element; pass B times that to let the B independent sorts run side by side */
size_t rd_sorted_foreground_workspace_bytes(long N, long k);
/* cls_score (B,N) [logits when apply_sigmoid != 0], bbox_delta (B,N,D), pc (B,N,3), mask (B,N) ->
 * sorted_fg_score (B,k), sorted_fg_bbox_delta (B,k,D), sorted_fg_pc (B,k,3); optional sorted_idx (B,k) int32.
 * Order: score*mask descending, ties by flat index ascending.  Requires N >= k (get_sorted_foreground.py:65).
 * (2k <= N: only the keys up to the k-th one's 12-bit bin are sorted -- same result, see k_sort.h.) */
int rd_sorted_foreground(const float* cls_score, const float* bbox_delta, const float* pc, const float* mask,
                         int B, long N, long k, int D, int apply_sigmoid, float* out_score, float* out_delta,
                         float* out_pc, int* out_idx, void* ws, size_t ws_bytes, void* stream);

/* bbox_delta (B,N,box_type) + pc (B,N,3) -> (B,N,10); box_type 8 (is_bin=0) or 7 (is_bin=1)
 * (shape checks of decode_3d_bbox.cc:30-49). */
int rd_decode3d_bbox(const float* bbox_delta, const float* pc, float* out, int B, long N, int box_type,
                     int is_bin, void* stream);

/* scores (n), boxes10 (n,10) -> dets (K,12) rows [8 corners, yaw, bottom, height, score] for score > min_score,
 * input order preserved; *d_count (device int) = K.  dets must hold n rows. */
size_t rd_score_filter_workspace_bytes(long n);
int rd_score_filter_dets(const float* scores, const float* boxes10, long n, float min_score, float* dets,
                         int* d_count, void* ws, size_t ws_bytes, void* stream);
/* B frames in one set of launches: frame b reads scores + b*scores_bstride, boxes10 + b*boxes_bstride, writes
 * dets + b*dets_bstride (strides in floats) and d_count[b]; ws_bytes >= B * rd_score_filter_workspace_bytes(n). */
int rd_score_filter_dets_batched(const float* scores, long scores_bstride, const float* boxes10, long boxes_bstride,
                                 long n, float min_score, float* dets, long dets_bstride, int* d_count, void* ws,
                                 size_t ws_bytes, int B, void* stream);

/* Weighted NMS.  dets (Kcap,12) device; the number of valid rows is *d_count when d_count != NULL, else Kcap (rows past
 * the count are ignored; a count above Kcap is clamped -- callers size Kcap from the data, up to RD_WNMS_MAX_K = more than
 * the reference's pre_nms_top_n of 50000 rows).
 * order: device int32 (Kcap) processing order (sorted positions -> row index).  NULL = the library orders on the device:
 *   tie_order RD_TIE_REFERENCE  the order std::sort gives point4_wnms_4c (nms.h:786-792; unstable: tied scores come out in
 *                               libstdc++ introsort order, replayed on the device; rows that arrive strictly sorted -- the
 *                               pipeline's case without ties -- cost one pass);
 *   tie_order RD_TIE_STABLE     score descending, ties by row index ascending (single frame only).
 * hash_scale: cell size of the reference's BBoxHash prefilter (nms.h:252-307, 459, 470, 501-507; tools/test.py:216 passes
 * 100): a pair of boxes is compared only when their cell ranges share a key, exactly like the reference.  <= 0: no
 * prefilter.  Outputs: out_dets (Kcap,12), keep (Kcap) row indices into dets in processing order, *d_nkeep = M. */
#define RD_WNMS_MAX_K 65536
#define RD_TIE_STABLE 0
#define RD_TIE_REFERENCE 1
/* Diagnostic bits that may be OR-ed into tie_order (per call; the RESULT never depends on them -- they select code paths so that tests
 * can compare them bit for bit; until round 5 these were environment variables of the library):
 *   RD_WNMS_DIAG_NO_SKIP        clip every pair the reference clips (no rejection test, see rd_wnms_pair_skippable)
 *   RD_WNMS_DIAG_TILE_W(n)      scan column chunks of n 64-row words (1..255): the column-chunked scan of K > 16 384 at any K
 *   RD_WNMS_DIAG_MERGE_LDS(n)   merge neighbourhood list of n entries in LDS (4..127): the global-scratch merge path at small K */
#define RD_WNMS_DIAG_NO_SKIP 0x100
#define RD_WNMS_DIAG_TILE_W(n) (((n) & 0xff) << 16)
#define RD_WNMS_DIAG_MERGE_LDS(n) (((n) & 0x7f) << 24)
size_t rd_wnms_workspace_bytes(int Kcap);
int rd_wnms_4c(const float* dets, int Kcap, const int* d_count, const int* order, int tie_order, float thresh,
               float thresh_vote, int is3d, int hash_scale, float* out_dets, int* keep, int* d_nkeep, void* ws,
               size_t ws_bytes, void* stream);
/* B independent frames in one set of launches (the per-frame greedy scan is one latency-bound wavefront: B of them
 * run side by side instead of back to back).  Frame b uses dets + b*dets_bstride, order + b*order_bstride (stride 0 =
 * one order shared by all frames; with order == NULL and B > 1 tie_order must be RD_TIE_REFERENCE), d_count[b],
 * out_dets + b*out_bstride, keep + b*keep_bstride, d_nkeep[b]; ws_bytes >= B * rd_wnms_workspace_bytes(Kcap).
 * Strides in elements. */
int rd_wnms_4c_batched(const float* dets, long dets_bstride, int Kcap, const int* d_count, const int* order,
                       long order_bstride, int tie_order, float thresh, float thresh_vote, int is3d, int hash_scale,
                       float* out_dets, long out_bstride, int* keep, long keep_bstride, int* d_nkeep, void* ws,
                       size_t ws_bytes, int B, void* stream);
/* OverlapChecker::single_overlap (nms.h:195-249) of n row pairs: out[i] = overlap(dets_a[i], dets_b[i]) -- BEV IoU of the two
 * 4-corner boxes, or volume IoU with is3d (rows are (n,12) dets rows).  The measure the weighted NMS thresholds. */
int rd_single_overlap(const float* dets_a, const float* dets_b, long n, int is3d, float* out, void* stream);
/* The pair kernel's rejection test on n independent row pairs (dets rows as in rd_wnms_4c): out[i] = 1 if rd_wnms_4c would NOT run
 * the polygon clip on (dets_a[i], dets_b[i]) because the reference's value cannot reach a threshold >= 1e-3 -- both boxes rectangles
 * with 0.2 .. 25 m edges and |coordinates| <= 200 m, bounding rectangles > 0.01 m apart, edge directions >= 0.01 rad apart mod 90
 * degrees (characterised on the compiled reference: oracle/ref_overlap_study.cpp; nms.h:96-149,195-249 has no empty-intersection
 * test).  Test / characterisation aid: results of rd_wnms_4c do not depend on it. */
int rd_wnms_pair_skippable(const float* dets_a, const float* dets_b, long n, unsigned char* out, void* stream);

/* out[i] = atan2f(y[i], x[i]) as the weighted NMS computes its edge angles (nms.h:71) and the score filter its yaw column: the C
 * library's float routine -- glibc's fdlibm algorithm restated operation by operation on the device (not the device math library's
 * atan2f, which differs from it in the last bit).  Test / characterisation aid: bit-equal to the host's atan2f (tests/test_kernels.py). */
int rd_edge_atan2f(const float* y, const float* x, long n, float* out, void* stream);

/* HOST: the reference's own ordering (std::sort, score descending, unstable) for dets_host (K,12). */
int rd_wnms_order_host(const float* dets_host, int K, int* order_host);

/* dets12 (M,12) -> (M,8) [cx,cy,cz,l,w,h,heading,score]; count from *d_count when not NULL. */
int rd_dets12_to_8(const float* dets12, int Mcap, const int* d_count, float* out8, void* stream);
int rd_dets12_to_8_batched(const float* dets12, long dets12_bstride, int Mcap, const int* d_count, float* out8,
                           long out8_bstride, int B, void* stream);

/* 8-point rotated IoU: boxes1 (n1,8) x boxes2 (n2,8) -> ious (n1,n2). */
int rd_rotated_iou_8pt(const float* boxes1, const float* boxes2, float* ious, long n1, long n2, void* stream);
/* Custom op 'batch_rotated_iou', iou_type 'bev' (operator_py/batch_rotated_iou.py:11-49, shapes :69-91): proposal
 * (B, N, p_stride >= 8: the 8 BEV corner coordinates lead each row; the op's rows are 10 wide), gt_bbox (B, n_gt <= 256, 8)
 * -> iou_map (B, N) = max over the frame's GT boxes of the 8-point rotated IoU after NaN / Inf / > 1 / < 0 -> 0.
 * argmax (B, N) int32, optional (NULL): index of the first GT box reaching that maximum (0 when every IoU is 0). */
int rd_batch_rotated_iou(const float* proposal, int p_stride, const float* gt_bbox, float* iou_map, int* argmax, int B, long N,
                         int n_gt, void* stream);
/* The same op with iou_type '3d' (operator_py/batch_rotated_iou.py:17-18,36-39,51-68; _contrib_RotatedIOU on 7-dim boxes,
 * operator_cxx/contrib/rotated_iou-inl.h:495-522): proposal (B, N, p_stride >= 10: 8 BEV corners + z0, z1) is converted to
 * [cx, cy, cz, length, width, height, yaw] (to_box_type_7), the yaw of proposals AND of gt_bbox7 (B, n_gt <= 256, 7) is negated, and
 * iou_map (B, N) = max over the frame's GT boxes of the VOLUME IoU after the same cleaning.  argmax as above.
 * rd_rotated_iou_7: the 7-dim IoU matrix itself, boxes [x, y, z, w, l, h, angle] as the reference op takes them. */
int rd_batch_rotated_iou_3d(const float* proposal, int p_stride, const float* gt_bbox7, float* iou_map, int* argmax, int B, long N,
                            int n_gt, void* stream);
int rd_rotated_iou_7(const float* boxes1, const float* boxes2, float* ious, long n1, long n2, void* stream);
/* one frame of the above without argmax.  proposals (n, p_stride>=8) */
int rd_batch_max_iou(const float* proposals, int p_stride, const float* gt8, float* out, long n, int n_gt,
                     void* stream);

/* ---- greedy 3-D NMS: _contrib_NMS3D(boxes, iou_thres, max_keep, normal_iou=False) ----
 * operator_cxx/contrib/nms_3d.cc:22-68 (shapes), nms_3d.cu:342-378 (overlap measures), :380-464 (mask + keep loop),
 * :466-534 (forward).  boxes (B,N,10) float32 = 4 BEV corners (x,y) + z_low, z_high, ALREADY sorted by score descending
 * (head/builder.py:519-534 feeds the decoded boxes of get_sorted_foreground).  Box i is kept when no kept box before it
 * has overlap(kept, i) > iou_thres; overlap = intersection volume / union volume (polygon clip in BEV x height overlap),
 * or the axis-aligned 2-D IoU of (x1,y1,x2,y2) = boxes[..., 0:4] when normal_iou != 0.  Stops after max_keep rows.
 * keep_idx (B,max_keep) int32 padded with -1; bbox_after_nms (B,max_keep,10) padded with 0.
 * Unlike the reference (N x N/64 words per frame) the mask is held for 1024 rows at a time. */
size_t rd_nms3d_workspace_bytes(long N, int B);
/* tools/test.py:193-196 (the `not pTest.nms.wnms` branch): out (B,max_keep) = score[b][keep_idx[b][i]], -inf where
 * keep_idx is -1, so that rd_score_filter_dets_batched on (out, bbox_after_nms) yields the reference's final rows. */
int rd_gather_keep_scores(const float* score, long score_bstride, int k, const int* keep_idx, int max_keep, float* out, int B,
                          void* stream);
int rd_nms3d(const float* boxes, int B, long N, float iou_thres, int max_keep, int normal_iou, int* keep_idx,
             float* bbox_after_nms, void* ws, size_t ws_bytes, void* stream);

/* ---- training-target assignment: processing_cxx.assign3D_v2 / get_point_num (pybinding.cpp:9-10) ----
 * assigner.h:11-85.  For every point (pc (N,3), row-major) the index of the FIRST ground-truth box that contains it, or -1.
 * bbox (M,24) = 8 corners x (x,y,z): A B C D bottom face, E.. top face; bbox_center (M,3); bbox_radius (M) is compared with
 * the SQUARED centre distance, and so is max_dist (the reference does both, :47-51); mask (N) < 0.5 or is_in_nlz (N) > 0
 * skips the point; the six limits are the caller's axis-aligned bounds of all boxes (input.py:303-308).  out (N) int32.
 * M <= 1024 (boxes are held in LDS). */
int rd_assign3d_v2(const float* pc, const float* bbox, const float* bbox_center, const float* bbox_radius, const float* mask,
                   const float* is_in_nlz, long N, int M, float max_x, float min_x, float max_y, float min_y, float max_z,
                   float min_z, float max_dist, int* out, void* stream);
/* assigner.h:87-109.  bbox_inds (N) float32 box index per point (negative = none); out (N) float32 = number of points that
 * share the point's box, -1 for points without a box.  Indices >= 500 (MAX_BOX_NUM, :92) are outside the reference's
 * contract (it writes out of bounds); here such points get -1. */
size_t rd_get_point_num_workspace_bytes(void);
int rd_get_point_num(const float* bbox_inds, long N, float* out, void* ws, size_t ws_bytes, void* stream);

/* ---- test-time input transform chain on the device (the step before the path) ----
 * rangedet/core/input.py:14-42,89-229,522-624 (LoadRecord, ProcessMissValue, SepAndClipData, GetUnnormalizedRange,
 * NormData, GetCoordinates, CombineData, PadData, TransposeData, GenerateFPNTarget, TransAndReshape) fused into one
 * kernel: raw record arrays in, the graph's named float32 tensors out.
 * Channel order of clip/mean/sd: range, intensity, elongation, x, y, z, inclination, azimuth (config:269-282; azimuth is
 * not clipped).  sd = sqrt(var) as float.  Level l uses stride 2^l and keeps range in [interval_lo[l], interval_hi[l]). */
typedef struct {
  float clip_lo[7], clip_hi[7];
  float mean[8], sd[8];
  float interval_lo[3], interval_hi[3];
} rd_input_norm_t;
/* range_image (B,H,W,4), pc_vehicle_frame (B,H,W,3), inclination (B,H)  ->  input_data (B,8,Hp,Wp), coord_s1 (B,3,Hp,Wp),
 * pc_vehicle_frame_s{1,2,4} (B,Hp*Wp/s,3), range_image_mask_s{1,2,4} (B,Hp*Wp/s); Hp >= H, Wp >= W, Wp % 4 == 0.
 * norm_host is read on the host at call time. */
int rd_input_transform(const float* range_image, const float* pc_vehicle_frame, const float* inclination,
                       const rd_input_norm_t* norm_host, int B, int H, int W, int Hp, int Wp, float* input_data,
                       float* coord_s1, float* pc_s1, float* pc_s2, float* pc_s4, float* mask_s1, float* mask_s2,
                       float* mask_s4, void* stream);

/* ---- per-kernel timing (HIP events on the launch stream; used by bench.py's roofline block) --------- */
#define RD_PROF_CONV 0
#define RD_PROF_META 1
#define RD_PROF_HEAD_OUT 2
#define RD_PROF_SORT 3
#define RD_PROF_DECODE 4
#define RD_PROF_WNMS 5
#define RD_PROF_LAYOUT 6
#define RD_PROF_CONV3 7 /* the persistent 3x3 stride-1 bf16 kernel (its launches are NOT counted in RD_PROF_CONV) */
#define RD_PROF_BLOCK 8 /* the fused BasicBlock kernel (rd_block64_bn_act) */
#define RD_PROF_NKINDS 9
int rd_prof_enable(int on);
int rd_prof_reset(void);
/* synchronises the recorded events; total_ms / launches per kind */
int rd_prof_get(int kind, double* total_ms, long* launches);

#ifdef __cplusplus
}
#endif
#endif /* RANGEDET_HIP_H_ */
