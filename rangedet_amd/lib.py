"""ctypes binding of librangedet_hip.so (include/rangedet_hip.h).

The product path: ``get_lib()`` opens ``rangedet_amd/librangedet_hip.so`` -- built by ``rangedet_amd.build`` with
hipcc for gfx950 -- and raises ``RuntimeError`` when it is missing.  There is no CPU fallback of any kind.
(``Lib(path)`` takes an explicit path only so that the CPU-only test tier can bind the hipemu build of the same
sources, tests/emu; nothing in this package ever does that.)
"""
import ctypes
import os

import numpy as np

RD_OK, RD_EINVAL, RD_ESHAPE, RD_EWORKSPACE, RD_EHIP = 0, -1, -2, -3, -4
RD_F32, RD_BF16, RD_F16 = 0, 1, 2
H16 = (RD_BF16, RD_F16)     # the two 16-bit element types (same layouts)
RD_RELU_PRE, RD_ADD, RD_RELU_POST, RD_SCALE_FOLDED, RD_MFMA16 = 1, 2, 4, 8, 16
RD_WNMS_MAX_K = 65536
# diagnostic bits of rd_wnms_4c's tie_order (include/rangedet_hip.h): per-call test aids, the result never depends on them
RD_WNMS_DIAG_NO_SKIP = 0x100


def RD_WNMS_DIAG_TILE_W(n):
    return (n & 0xff) << 16


def RD_WNMS_DIAG_MERGE_LDS(n):
    return (n & 0x7f) << 24

RD_TIE_STABLE, RD_TIE_REFERENCE = 0, 1
PROF_KINDS = {"conv": 0, "meta": 1, "head_out": 2, "sort": 3, "decode": 4, "wnms": 5, "layout": 6, "conv3": 7, "block": 8}

_HERE = os.path.dirname(os.path.abspath(__file__))
# RANGEDET_HIP_LIB: another build of the same library (A/B timing of two builds on one box); never a CPU substitute
DEFAULT_PATH = os.environ.get("RANGEDET_HIP_LIB") or os.path.join(_HERE, "librangedet_hip.so")

c_int, c_long, c_float, c_size_t, c_void_p = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p

# name -> (restype, argtypes); every symbol include/rangedet_hip.h declares
SIGNATURES = {
    "rd_version": (c_int, []),
    "rd_last_error_string": (ctypes.c_char_p, []),
    "rd_nchw_to_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rd_nhwc_to_nchw": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rd_copy_rows": (c_int, [c_void_p, c_long, c_void_p, c_long, c_long, c_long, c_int, c_void_p]),
    "rd_conv_packed_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "rd_pack_conv_weight_host": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rd_deconv_phase_taps": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "rd_pack_deconv_weight_host": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rd_pack_deconv_weight_folded_host": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rd_conv2d_bn_act": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                 c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rd_conv3x3_ex_packed_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "rd_pack_conv3x3_ex_host": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rd_pack_conv3x3_m16_host": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "rd_conv3x3_mfma16_ok": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "rd_conv1x1_sc_packed_bytes": (c_size_t, [c_int, c_int]),
    "rd_pack_conv1x1_sc_host": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "rd_conv3x3_bn_act_ex": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                                     c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_int, c_int, c_void_p]),
    "rd_deconv2d_bn_act": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                   c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_int, c_void_p]),
    "rd_conv3x3_cat_packed_bytes": (c_size_t, [c_int, c_int, c_int]),
    "rd_pack_conv3x3_cat_host": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rd_conv3x3_bn_act_cat": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rd_deconv2d_all_phases_ok": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "rd_deconv2d_bn_act_all": (c_int, [c_void_p, c_int, c_int, c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                       c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_int, c_void_p]),
    "rd_deconv2d_phase_pairs_ok": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "rd_pack_deconv_phase_pair_host": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "rd_deconv2d_bn_act_pairs": (c_int, [c_void_p, c_int, c_int, c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                         c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                         c_int, c_void_p]),
    "rd_block64_packed_bytes": (c_size_t, [c_int]),
    "rd_pack_block64_host": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "rd_pack_block64_m16_host": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "rd_pack_conv1x1_sc_m16_host": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "rd_block64_m16_bn_act": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                      c_int, c_int, c_void_p]),
    "rd_block64_bn_act": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_void_p]),
    "rd_head_packed_bytes": (c_size_t, []),
    "rd_head_m16_packed_bytes": (c_size_t, []),
    "rd_pack_head_weight_host": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "rd_pack_head_weight_m16_host": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "rd_conv2d_bn_act_head_out": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                          c_void_p, c_void_p, c_void_p, c_long, c_long, c_int, c_int, c_void_p]),
    "rd_conv3x3_bn_act_pair": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int] * 2 +
                               [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rd_conv2d_bn_act_head_out_pair": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int] * 2 +
                                       [c_int, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rd_head_out": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_long, c_long, c_int, c_int, c_int,
                            c_int, c_int, c_int, c_void_p]),
    "rd_meta_packed_bytes": (c_size_t, [c_int]),
    "rd_pack_meta_host": (c_int, [c_void_p] * 9 + [c_int, c_void_p]),
    "rd_meta_kernel_fwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                   c_int, c_int, c_void_p]),
    "rd_sorted_foreground_workspace_bytes": (c_size_t, [c_long, c_long]),
    "rd_sorted_foreground": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long, c_long, c_int, c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rd_decode3d_bbox": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_void_p]),
    "rd_score_filter_workspace_bytes": (c_size_t, [c_long]),
    "rd_score_filter_dets": (c_int, [c_void_p, c_void_p, c_long, c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rd_score_filter_dets_batched": (c_int, [c_void_p, c_long, c_void_p, c_long, c_long, c_float, c_void_p, c_long, c_void_p,
                                             c_void_p, c_size_t, c_int, c_void_p]),
    "rd_wnms_workspace_bytes": (c_size_t, [c_int]),
    "rd_wnms_4c_batched": (c_int, [c_void_p, c_long, c_int, c_void_p, c_void_p, c_long, c_int, c_float, c_float, c_int, c_int,
                                   c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "rd_dets12_to_8_batched": (c_int, [c_void_p, c_long, c_int, c_void_p, c_void_p, c_long, c_int, c_void_p]),
    "rd_wnms_4c": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_float, c_float, c_int, c_int, c_void_p, c_void_p,
                           c_void_p, c_void_p, c_size_t, c_void_p]),
    "rd_single_overlap": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p]),
    "rd_wnms_order_host": (c_int, [c_void_p, c_int, c_void_p]),
    "rd_edge_atan2f": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    "rd_wnms_pair_skippable": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    "rd_dets12_to_8": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "rd_rotated_iou_8pt": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_long, c_void_p]),
    "rd_batch_rotated_iou": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_void_p]),
    "rd_batch_rotated_iou_3d": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_void_p]),
    "rd_rotated_iou_7": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_long, c_void_p]),
    "rd_batch_max_iou": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_long, c_int, c_void_p]),
    "rd_gather_keep_scores": (c_int, [c_void_p, c_long, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "rd_nms3d_workspace_bytes": (c_size_t, [c_long, c_int]),
    "rd_nms3d": (c_int, [c_void_p, c_int, c_long, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rd_assign3d_v2": (c_int, [c_void_p] * 6 + [c_long, c_int] + [c_float] * 7 + [c_void_p, c_void_p]),
    "rd_get_point_num_workspace_bytes": (c_size_t, []),
    "rd_get_point_num": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rd_input_transform": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int] + [c_void_p] * 8 + [c_void_p]),
    "rd_prof_enable": (c_int, [c_int]),
    "rd_prof_reset": (c_int, []),
    "rd_prof_get": (c_int, [c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_long)]),
}


class RangeDetError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("librangedet_hip error %d: %s" % (code, msg))
        self.code = code


def ptr(x):
    """Raw address of a numpy array, a torch tensor, an int or None."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    raise TypeError("cannot take the address of %r" % type(x))


class Lib:
    def __init__(self, path=DEFAULT_PATH):
        if not os.path.exists(path):
            raise RuntimeError(
                "%s not found: the HIP extension is not built (run `python -m rangedet_amd.build`). "
                "There is no CPU fallback." % path)
        self.path = path
        # PyTorch-ROCm wheels bundle their own HIP runtime.  If this library is dlopen'ed first it pulls in /opt/rocm's
        # copy, torch then loads its own, and the process ends up with two runtimes -- the second one sees no device
        # ("no ROCm-capable device is detected").  Loading torch first makes both resolve to the same runtime.
        if os.path.basename(path).startswith("librangedet_hip"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        self.cdll = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.cdll, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        self.rd_last_error_string = self.cdll.rd_last_error_string

    def call(self, name, *args):
        """Invoke a status-returning entry point; raise RangeDetError on a negative status."""
        rc = getattr(self.cdll, name)(*args)
        if rc < 0:
            raise RangeDetError(rc, self.cdll.rd_last_error_string().decode())
        return rc

    def raw(self, name):
        return getattr(self.cdll, name)

    # ---- host-side packers (numpy in, numpy out) ---------------------------------------------------------
    def pack_conv_weight(self, w_oihw, dtype):
        w = np.ascontiguousarray(w_oihw, dtype=np.float32)
        cout, cin, kh, kw = w.shape
        out = np.zeros(self.cdll.rd_conv_packed_bytes(kh * kw, cin, cout, dtype), dtype=np.uint8)
        self.call("rd_pack_conv_weight_host", w.ctypes.data, cout, cin, kh, kw, dtype, out.ctypes.data)
        return out

    def pack_conv3x3_ex(self, w_oihw, stride_w, x_cstride, fold_scale=None, dtype=RD_BF16):
        w = np.ascontiguousarray(w_oihw, dtype=np.float32)
        cout, cin = w.shape[:2]
        assert w.shape[2:] == (3, 3)
        fs = None if fold_scale is None else np.ascontiguousarray(fold_scale, dtype=np.float32)
        out = np.zeros(self.cdll.rd_conv3x3_ex_packed_bytes(cin, cout, stride_w, x_cstride), dtype=np.uint8)
        self.call("rd_pack_conv3x3_ex_host", w.ctypes.data, None if fs is None else fs.ctypes.data, cout, cin, stride_w,
                  x_cstride, dtype, out.ctypes.data)
        return out

    def pack_conv3x3_m16(self, w_oihw, fold_scale, dtype=RD_BF16):
        """weights of an RD_MFMA16 launch of rd_conv3x3_bn_act_ex (stride 1, cout 128, cin a multiple of 32, folded scales)"""
        w = np.ascontiguousarray(w_oihw, dtype=np.float32)
        cout, cin = w.shape[:2]
        assert w.shape[2:] == (3, 3)
        fs = np.ascontiguousarray(fold_scale, dtype=np.float32)
        out = np.zeros(self.cdll.rd_conv3x3_ex_packed_bytes(cin, cout, 1, cin), dtype=np.uint8)
        self.call("rd_pack_conv3x3_m16_host", w.ctypes.data, fs.ctypes.data, cout, cin, dtype, out.ctypes.data)
        return out

    def pack_conv3x3_cat(self, w_oihw, cin1, cin2, fold_scale=None, dtype=RD_BF16):
        """weights of rd_conv3x3_bn_act_cat: input channels [cin1 of x1 | cin2 of x2]"""
        w = np.ascontiguousarray(w_oihw, dtype=np.float32)
        cout = w.shape[0]
        assert w.shape[1:] == (cin1 + cin2, 3, 3)
        fs = None if fold_scale is None else np.ascontiguousarray(fold_scale, dtype=np.float32)
        out = np.zeros(self.cdll.rd_conv3x3_cat_packed_bytes(cin1, cin2, cout), dtype=np.uint8)
        self.call("rd_pack_conv3x3_cat_host", w.ctypes.data, None if fs is None else fs.ctypes.data, cout, cin1, cin2, dtype, out.ctypes.data)
        return out

    def pack_conv1x1_sc(self, w_oi, fold_scale=None, dtype=RD_BF16):
        w = np.ascontiguousarray(w_oi, dtype=np.float32).reshape(w_oi.shape[0], -1)
        cout, cin = w.shape
        fs = None if fold_scale is None else np.ascontiguousarray(fold_scale, dtype=np.float32)
        out = np.zeros(self.cdll.rd_conv1x1_sc_packed_bytes(cin, cout), dtype=np.uint8)
        self.call("rd_pack_conv1x1_sc_host", w.ctypes.data, None if fs is None else fs.ctypes.data, cout, cin, dtype, out.ctypes.data)
        return out

    def pack_block64(self, w1_oihw, scale1, w2_oihw, scale2, dtype=RD_BF16, m16=False):
        """weights of rd_block64_bn_act: the two 3x3 convs of a BasicBlock, (64, cin, 3, 3) with cin = 64 or <= 16 and (64, 64, 3, 3), with
        their BatchNorm scales folded in.  m16: the image of rd_block64_m16_bn_act (cin = 64)"""
        w1, w2 = (np.ascontiguousarray(w, dtype=np.float32) for w in (w1_oihw, w2_oihw))
        s1, s2 = (np.ascontiguousarray(v, dtype=np.float32) for v in (scale1, scale2))
        cin = w1.shape[1]
        assert w1.shape == (64, cin, 3, 3) and w2.shape == (64, 64, 3, 3) and s1.shape == (64,) and s2.shape == (64,)
        out = np.zeros(self.cdll.rd_block64_packed_bytes(cin), dtype=np.uint8)
        if m16:
            assert cin == 64
            self.call("rd_pack_block64_m16_host", w1.ctypes.data, s1.ctypes.data, w2.ctypes.data, s2.ctypes.data, dtype, out.ctypes.data)
        else:
            self.call("rd_pack_block64_host", w1.ctypes.data, s1.ctypes.data, w2.ctypes.data, s2.ctypes.data, cin, dtype, out.ctypes.data)
        return out

    def pack_conv1x1_sc_m16(self, w_oi, fold_scale, dtype=RD_BF16):
        """projection shortcut 64 -> 64 of rd_block64_m16_bn_act"""
        w = np.ascontiguousarray(w_oi, dtype=np.float32).reshape(64, 64)
        fs = np.ascontiguousarray(fold_scale, dtype=np.float32)
        out = np.zeros(self.cdll.rd_conv1x1_sc_packed_bytes(64, 64), dtype=np.uint8)
        self.call("rd_pack_conv1x1_sc_m16_host", w.ctypes.data, fs.ctypes.data, dtype, out.ctypes.data)
        return out

    def pack_deconv_weight(self, w_iohw, stride_w, pad_w, phase, dtype, fold_scale=None):
        w = np.ascontiguousarray(w_iohw, dtype=np.float32)
        cin, cout, kh, kw = w.shape
        nt = self.cdll.rd_deconv_phase_taps(kh, kw, stride_w, pad_w, phase)
        if nt < 0:
            raise RangeDetError(nt, "deconv phase taps")
        out = np.zeros(self.cdll.rd_conv_packed_bytes(nt, cin, cout, dtype), dtype=np.uint8)
        fs = None if fold_scale is None else np.ascontiguousarray(fold_scale, dtype=np.float32)
        self.call("rd_pack_deconv_weight_folded_host", w.ctypes.data, None if fs is None else fs.ctypes.data, cin, cout, kh, kw,
                  stride_w, pad_w, phase, dtype, out.ctypes.data)
        return out

    def pack_deconv_phase_pair(self, img_a, img_b, cin, dtype):
        """Two phase images of pack_deconv_weight (cout 64, folded scale) -> the pair image of rd_deconv2d_bn_act_pairs."""
        assert img_a.nbytes == img_b.nbytes
        out = np.empty(2 * img_a.nbytes, np.uint8)
        a, b = np.ascontiguousarray(img_a), np.ascontiguousarray(img_b)
        self.call("rd_pack_deconv_phase_pair_host", a.ctypes.data, b.ctypes.data, cin, dtype, out.ctypes.data)
        return out

    def pack_head_weight(self, w, dtype=RD_BF16, m16=False):
        """m16: the image an RD_MFMA16 launch of rd_conv2d_bn_act_head_out reads"""
        w = np.ascontiguousarray(w, dtype=np.float32)
        out = np.zeros(self.cdll.rd_head_m16_packed_bytes() if m16 else self.cdll.rd_head_packed_bytes(), dtype=np.uint8)
        self.call("rd_pack_head_weight_m16_host" if m16 else "rd_pack_head_weight_host", w.ctypes.data, w.shape[0], w.shape[1], dtype, out.ctypes.data)
        return out

    def pack_meta(self, w0, b0, w1, b1, s1, t1, agg, s2, t2, dtype):
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (w0, b0, w1, b1, s1, t1, agg, s2, t2)]
        assert arrs[0].shape == (32, 3) and arrs[2].shape == (64, 32) and arrs[6].shape == (64, 576)
        out = np.zeros(self.cdll.rd_meta_packed_bytes(dtype), dtype=np.uint8)
        self.call("rd_pack_meta_host", *[a.ctypes.data for a in arrs], dtype, out.ctypes.data)
        return out

    def wnms_order_host(self, dets):
        d = np.ascontiguousarray(dets, dtype=np.float32).reshape(-1, 12)
        order = np.empty(d.shape[0], dtype=np.int32)
        self.call("rd_wnms_order_host", d.ctypes.data, d.shape[0], order.ctypes.data)
        return order

    def prof(self):
        out = {}
        ms, n = ctypes.c_double(), c_long()
        for k, v in PROF_KINDS.items():
            self.call("rd_prof_get", v, ctypes.byref(ms), ctypes.byref(n))
            out[k] = (ms.value, n.value)
        return out


_LIB = None


def get_lib():
    """The product library.  Raises RuntimeError if librangedet_hip.so has not been built."""
    global _LIB
    if _LIB is None:
        _LIB = Lib(DEFAULT_PATH)
    return _LIB
