"""Drop-in for the reference's pybind11 module ``processing_cxx`` 
(operator_cxx/src_cxx/pybinding.cpp:6-11): ``wnms_4c(dets, thresh, thresh_vote, _3D, hash_scale) -> (list, list)``
with positional arguments only, ``assign3D_v2``, ``get_point_num``; executed by the HIP kernels of librangedet_hip.so.

Ordering: the reference sorts with std::sort (unstable, nms.h:786-792); the library replays that sort on the device
(RD_TIE_REFERENCE) so tied scores are processed exactly as in the reference.  ``hash_scale`` is the cell size of the
reference's BBoxHash prefilter (nms.h:252-307): boxes without a common cell are never compared, here as there.
``assign3D_v2`` / ``get_point_num`` (training-target generation, assigner.h) run on the GPU as well.
"""
import numpy as np

from . import lib as rdlib
from .runtime import TorchAllocator

_STATE = {}


def _ctx():
    if not _STATE:
        _STATE["lib"] = rdlib.get_lib()
        _STATE["alloc"] = TorchAllocator()
    return _STATE["lib"], _STATE["alloc"]


def wnms_4c(dets, thresh, thresh_vote, _3D, hash_scale):
    d = np.ascontiguousarray(np.asarray(dets, dtype=np.float32)).reshape(-1, 12)
    K = d.shape[0]
    if K == 0:
        return [], []  # nms.h:463-466
    if K > rdlib.RD_WNMS_MAX_K:
        raise rdlib.RangeDetError(rdlib.RD_ESHAPE, "wnms_4c: %d boxes exceed RD_WNMS_MAX_K" % K)
    L, A = _ctx()
    dd = A.upload(d)
    nb = L.raw("rd_wnms_workspace_bytes")(K)
    ws, out, keep, nk = A.alloc(nb), A.alloc(K * 48), A.alloc(K * 4), A.alloc(16, zero=True)
    L.call("rd_wnms_4c", A.ptr(dd), K, None, None, rdlib.RD_TIE_REFERENCE, float(thresh), float(thresh_vote), int(bool(_3D)),
           int(hash_scale), A.ptr(out), A.ptr(keep), A.ptr(nk), A.ptr(ws), nb, A.stream)
    A.sync()
    M = int(A.to_numpy(A.view_i32(nk, (1,)))[0])
    rows = A.to_numpy(A.view_f32(out, (K, 12)))[:M]
    return rows.reshape(-1).tolist(), A.to_numpy(A.view_i32(keep, (K,)))[:M].tolist()


def assign3D_v2(pc, bbox, bbox_center, bbox_radius, mask, is_in_nlz, max_x_, min_x_, max_y_, min_y_, max_z_, min_z_,
                max_dist):
    """assigner.h:11-85: (N,3) points, (M,24) boxes, (M,3) centres, (M,1) radii, (N,1) mask, (N,1) no-label-zone flags
    -> (N,1) int32 index of the first containing box, -1 for none (same argument order as the pybind11 function)."""
    f = lambda x, w: np.ascontiguousarray(np.asarray(x, dtype=np.float32)).reshape(-1, w)  # noqa: E731
    p, b, c = f(pc, 3), f(bbox, 24), f(bbox_center, 3)
    r, m, z = f(bbox_radius, 1), f(mask, 1), f(is_in_nlz, 1)
    N, M = p.shape[0], b.shape[0]
    if not (c.shape[0] == M and r.shape[0] == M and m.shape[0] == N and z.shape[0] == N):
        raise rdlib.RangeDetError(rdlib.RD_ESHAPE, "assign3D_v2: inconsistent row counts")
    if N == 0:
        return np.zeros((0, 1), np.int32)
    L, A = _ctx()
    out = A.alloc(N * 4)
    dev = [A.upload(x) for x in (p, b, c, r, m, z)]          # held until the sync below
    L.call("rd_assign3d_v2", *[A.ptr(t) for t in dev], N, M, float(max_x_), float(min_x_), float(max_y_), float(min_y_),
           float(max_z_), float(min_z_), float(max_dist), A.ptr(out), A.stream)
    A.sync()
    return np.array(A.to_numpy(A.view_i32(out, (N, 1))))


def get_point_num(bbox_inds_each_pt):
    """assigner.h:87-109: (N,) or (N,1) float box index per point -> (N,1) float32 points-in-that-box, -1 for no box."""
    v = np.ascontiguousarray(np.asarray(bbox_inds_each_pt, dtype=np.float32)).reshape(-1)
    N = v.shape[0]
    if N == 0:
        return np.zeros((0, 1), np.float32)
    L, A = _ctx()
    nb = L.raw("rd_get_point_num_workspace_bytes")()
    ws, out = A.alloc(nb), A.alloc(N * 4)
    dv = A.upload(v)
    L.call("rd_get_point_num", A.ptr(dv), N, A.ptr(out), A.ptr(ws), nb, A.stream)
    A.sync()
    return np.array(A.to_numpy(A.view_f32(out, (N, 1))))
