"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

ctypes front-end of oracle/liboracle.so (C++ restatement, oracle/oracle.cpp) plus numpy restatements of
the host-side numpy steps of the reference:
  * bbox3d_10dim_to_11dim / score filter   tools/test.py:56-81,200-209
  * bbox3d_12dim_to_8dim                   tools/test.py:43-53
  * get_sorted_foreground                  operator_py/get_sorted_foreground.py:11-40
Citations are relative to /root/reference.
"""
import ctypes
import glob
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(with_ref=True):
    """(Re)build liboracle.so and, when /root/reference is present, oracle/_ref (the reference's own wnms)."""
    targets = ["all"]
    if with_ref and os.path.exists("/root/reference/operator_cxx/src_cxx/nms.h"):
        targets += ["ref", "study"]      # (study: the bulk characterisation tool of tools/nms_spurious_study.py)
    subprocess.check_call(["make", "-s", "-C", _HERE] + targets)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(with_ref=False)
        L = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int)
        L.orc_single_overlap.restype = ctypes.c_float
        L.orc_single_overlap.argtypes = [fp, fp, ctypes.c_int]
        L.orc_wnms_order.argtypes = [fp, ctypes.c_int, ip]
        L.orc_antiqsort_keys.argtypes = [ctypes.c_int, fp]
        L.orc_std_sort_depth_limit_hit.argtypes = [fp, ctypes.c_int]
        L.orc_std_sort_depth_limit_hit.restype = ctypes.c_int
        L.orc_wnms_4c.restype = ctypes.c_int
        L.orc_wnms_4c.argtypes = [fp, ip, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                  ctypes.c_int, fp, ip]
        L.orc_decode3d.argtypes = [fp, fp, fp, ctypes.c_long, ctypes.c_int, ctypes.c_int]
        L.orc_rotated_iou_8pt.argtypes = [fp, fp, fp, ctypes.c_long, ctypes.c_long]
        L.orc_rotated_iou_7.argtypes = [fp, fp, fp, ctypes.c_long, ctypes.c_long]
        L.orc_to_box_type_7.argtypes = [fp, fp, ctypes.c_long]
        L.orc_assign3d_v2.argtypes = [fp] * 6 + [ctypes.c_long, ctypes.c_int] + [ctypes.c_float] * 7 + [ip]
        L.orc_get_point_num.argtypes = [fp, ctypes.c_long, fp]
        L.orc_nms3d_overlap.argtypes = [fp, fp, fp, ctypes.c_long, ctypes.c_long, ctypes.c_int]
        L.orc_nms3d.argtypes = [fp, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_int, ctypes.c_int, ip, fp]
        _LIB = L
    return _LIB


def ref_module():
    """The reference's own wnms compiled from /root/reference (oracle/_ref), or None when not built."""
    import importlib.util
    hits = glob.glob(os.path.join(_HERE, "_ref", "processing_cxx_ref*.so"))
    if not hits:
        return None
    spec = importlib.util.spec_from_file_location("processing_cxx_ref", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def single_overlap(box1, box2, is3d=False):
    b1, p1 = _f(box1)
    b2, p2 = _f(box2)
    return float(lib().orc_single_overlap(p1, p2, int(is3d)))


def wnms_order(dets):
    d, p = _f(dets)
    K = d.shape[0]
    order = np.empty(K, dtype=np.int32)
    lib().orc_wnms_order(p, K, order.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return order


def antiqsort_keys(n):
    """n distinct scores on which the reference's std::sort (nms.h:791) degenerates into its heap-sort branch."""
    k = np.empty(n, dtype=np.float32)
    lib().orc_antiqsort_keys(n, k.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return k


def std_sort_depth_limit_hit(keys):
    k, p = _f(keys)
    return bool(lib().orc_std_sort_depth_limit_hit(p, k.shape[0]))


def wnms_4c(dets, thresh, thresh_vote, is3d=False, hash_scale=100, order=None):
    """Same contract as processing_cxx.wnms_4c (pybinding.cpp:8): returns (flat list M*12, keep list M)."""
    d, p = _f(dets)
    d = d.reshape(-1, 12)
    K = d.shape[0]
    if K == 0:
        return [], []
    if order is None:
        order = wnms_order(d)
    order = np.ascontiguousarray(order, dtype=np.int32)
    out = np.empty((K, 12), dtype=np.float32)
    keep = np.empty(K, dtype=np.int32)
    M = lib().orc_wnms_4c(p, order.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), K, thresh, thresh_vote,
                          int(is3d), int(hash_scale), out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                          keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return out[:M].reshape(-1).tolist(), keep[:M].tolist()


def decode3d(delta, pc, is_bin=False):
    """delta (B,N,8|7), pc (B,N,3) -> (B,N,10)  (decode_3d_bbox.cc:30-49 shapes)."""
    d, pd = _f(delta)
    c, pcp = _f(pc)
    assert d.ndim == 3 and c.ndim == 3 and c.shape[2] == 3 and d.shape[:2] == c.shape[:2]
    out = np.empty(d.shape[:2] + (10,), dtype=np.float32)
    lib().orc_decode3d(pd, pcp, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                       d.shape[0] * d.shape[1], d.shape[2], int(is_bin))
    return out


def rotated_iou_8pt(b1, b2):
    a, pa = _f(b1)
    b, pb = _f(b2)
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.float32)
    lib().orc_rotated_iou_8pt(pa, pb, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), a.shape[0], b.shape[0])
    return out


def nms3d_overlap(b1, b2, normal_iou=False):
    """nms_3d.cu:342-378: the pairwise measure NMS3D thresholds, boxes (n,10)."""
    a, pa = _f(b1)
    b, pb = _f(b2)
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.float32)
    lib().orc_nms3d_overlap(pa, pb, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), a.shape[0], b.shape[0], int(normal_iou))
    return out


def nms3d(boxes, iou_thres, max_keep, normal_iou=False):
    """_contrib_NMS3D (nms_3d.cu:380-534): boxes (B,N,10) sorted by score -> keep_idx (B,max_keep) int32 (-1 padded),
    bbox_after_nms (B,max_keep,10) (0 padded)."""
    bx, pb = _f(boxes)
    B, N = bx.shape[0], bx.shape[1]
    keep = np.empty((B, max_keep), dtype=np.int32)
    out = np.empty((B, max_keep, 10), dtype=np.float32)
    lib().orc_nms3d(pb, B, N, float(iou_thres), int(max_keep), int(normal_iou),
                    keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return keep, out


def assign3d_v2(pc, bbox, center, radius, mask, nlz, max_x, min_x, max_y, min_y, max_z, min_z, max_dist):
    """assigner.h:11-85 -> (N,) int32."""
    (p, pp), (b, pb), (c, pcn), (r, pr), (m, pm), (z, pz) = (_f(x) for x in (pc, bbox, center, radius, mask, nlz))
    N, M = p.reshape(-1, 3).shape[0], b.reshape(-1, 24).shape[0]
    out = np.empty(N, dtype=np.int32)
    lib().orc_assign3d_v2(pp, pb, pcn, pr, pm, pz, N, M, max_x, min_x, max_y, min_y, max_z, min_z, max_dist,
                          out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return out


def get_point_num(inds):
    """assigner.h:87-109 -> (N,) float32."""
    v, pv = _f(inds)
    out = np.empty(v.size, dtype=np.float32)
    lib().orc_get_point_num(pv, v.size, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def batch_max_iou(proposal8, gt8):
    """operator_py/batch_rotated_iou.py:33-49 ('bev'): clean NaN/Inf/>1/<0 to 0, max over GT."""
    m = rotated_iou_8pt(proposal8, gt8)
    m[np.isnan(m)] = 0
    m[np.isinf(m)] = 0
    m[m > 1.0] = 0
    m[m < 0] = 0
    return m.max(axis=1)


def rotated_iou_7(b1, b2):
    """_contrib_RotatedIOU on 7-dim boxes [x, y, z, w, l, h, angle] (rotated_iou-inl.h:495-522): volume IoU matrix."""
    a, pa = _f(b1)
    b, pb = _f(b2)
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.float32)
    lib().orc_rotated_iou_7(pa, pb, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), a.shape[0], b.shape[0])
    return out


def to_box_type_7(p10):
    """BatchRotatedIOU.to_box_type_7 (operator_py/batch_rotated_iou.py:51-68): (n,10) corner boxes -> (n,7)."""
    a, pa = _f(p10)
    out = np.empty((a.shape[0], 7), dtype=np.float32)
    lib().orc_to_box_type_7(pa, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), a.shape[0])
    return out


def batch_max_iou_3d(proposal10, gt7):
    """operator_py/batch_rotated_iou.py:17-18,36-49 ('3d'): proposals to 7-dim, the yaw of BOTH sides negated (:37-38), volume IoU,
    clean NaN/Inf/>1/<0 to 0, max over GT.  Returns (max, full matrix)."""
    r = to_box_type_7(proposal10)
    g = np.array(gt7, dtype=np.float32)
    r[:, 6] = -1 * r[:, 6]
    g[:, 6] = -1 * g[:, 6]
    m = rotated_iou_7(r, g)
    m[np.isnan(m)] = 0
    m[np.isinf(m)] = 0
    m[m > 1.0] = 0
    m[m < 0] = 0
    return m.max(axis=1), m


def bbox3d_10dim_to_11dim(b10):
    b10 = np.array(b10, dtype=np.float32)
    xy = b10[:, :8]
    bottom = b10[:, 8:9]
    top = b10[:, 9:10]
    yaw = np.arctan2(xy[:, 1] - xy[:, 3], xy[:, 0] - xy[:, 2])
    return np.concatenate([xy, yaw[:, None], bottom, top - bottom], axis=1)


def score_filter_to_dets(cls_score, bbox10, min_score):
    """tools/test.py:200-209 : (K,12) float32 rows [8 corners, yaw, bottom, height, score], order preserved."""
    fg = cls_score > min_score
    s = cls_score[fg]
    b = bbox10[fg]
    if b.shape[0] == 0:
        return np.zeros((0, 12), np.float32)
    return np.concatenate([bbox3d_10dim_to_11dim(b), s[:, None]], axis=1).astype(np.float32)


def bbox3d_12dim_to_8dim(b12):
    b12 = np.asarray(b12)
    cx = np.mean(b12[:, [0, 2, 4, 6]], axis=1)
    cy = np.mean(b12[:, [1, 3, 5, 7]], axis=1)
    z0 = b12[:, 9]
    h = b12[:, 10]
    cz = z0 + h / 2
    length = np.sqrt((b12[:, 2] - b12[:, 0]) ** 2 + (b12[:, 3] - b12[:, 1]) ** 2)
    width = np.sqrt((b12[:, 2] - b12[:, 4]) ** 2 + (b12[:, 3] - b12[:, 5]) ** 2)
    return np.stack([cx, cy, cz, length, width, h, b12[:, 8], b12[:, 11]], axis=1)


def get_sorted_foreground(cls_score, bbox_delta, pc, mask, num_fgs):
    """operator_py/get_sorted_foreground.py:11-40.  MXNet's topk/argsort tie order is third-party and
    unpinned; this restatement (and the build) use (score desc, flat index asc)."""
    s = (cls_score * mask).astype(np.float32)
    B, N = s.shape
    assert N >= num_fgs
    sc = np.empty((B, num_fgs), np.float32)
    dl = np.empty((B, num_fgs, bbox_delta.shape[2]), np.float32)
    pp = np.empty((B, num_fgs, 3), np.float32)
    idx = np.empty((B, num_fgs), np.int64)
    for b in range(B):
        o = np.argsort(-s[b], kind="stable")[:num_fgs]
        idx[b] = o
        sc[b] = s[b][o]
        dl[b] = bbox_delta[b][o]
        pp[b] = pc[b][o]
    return sc, dl, pp, idx
