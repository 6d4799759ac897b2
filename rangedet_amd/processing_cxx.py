"""Drop-in for the reference's pybind11 module ``processing_cxx`` on the inference path
(operator_cxx/src_cxx/pybinding.cpp:6-11): ``wnms_4c(dets, thresh, thresh_vote, _3D, hash_scale) -> (list, list)``
with positional arguments only, executed by the HIP kernels of librangedet_hip.so.

Ordering: the reference sorts with std::sort (unstable, nms.h:786-792); ``rd_wnms_order_host`` runs that same call on
the host so tied scores come out exactly as in the reference, everything else happens on the GPU.  ``hash_scale`` only
parameterises the reference's spatial prefilter, which never rejects a pair that could matter; it is accepted
and ignored.  ``assign3D_v2`` / ``get_point_num`` are training-time target generation and are out of scope.
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
    order = L.wnms_order_host(d)
    dd, od = A.upload(d), A.upload(order)
    nb = L.raw("rd_wnms_workspace_bytes")(K)
    ws, out, keep, nk = A.alloc(nb), A.alloc(K * 48), A.alloc(K * 4), A.alloc(16, zero=True)
    L.call("rd_wnms_4c", A.ptr(dd), K, None, A.ptr(od), float(thresh), float(thresh_vote), int(bool(_3D)), A.ptr(out),
           A.ptr(keep), A.ptr(nk), A.ptr(ws), nb, A.stream)
    A.sync()
    M = int(A.to_numpy(A.view_i32(nk, (1,)))[0])
    rows = A.to_numpy(A.view_f32(out, (K, 12)))[:M]
    return rows.reshape(-1).tolist(), A.to_numpy(A.view_i32(keep, (K,)))[:M].tolist()


def assign3D_v2(*a, **k):
    raise NotImplementedError("assign3D_v2 is training-time target generation (out of scope, DESIGN.md)")


get_point_num = assign3D_v2
