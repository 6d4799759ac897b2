"""Timing of rd_nms3d at the shipped configs' size (dev tool): python tools/nms3d_bench.py [N] [B] [max_keep]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from rangedet_amd import lib as R, synth  # noqa: E402
from rangedet_amd.runtime import TorchAllocator  # noqa: E402

N, B, mk = [int(v) for v in (sys.argv[1:4] + ["50000", "8", "200"][len(sys.argv) - 1:])]
L, A = R.get_lib(), TorchAllocator()
rep = 400
frames = []
for b in range(B):
    d = synth.cluster_dets(N // rep, rep, seed=3 + b, jitter=0.15)
    d = d[np.argsort(-d[:, 11], kind="stable")]
    frames.append(np.concatenate([d[:, :8], d[:, 9:10], d[:, 9:10] + d[:, 10:11]], axis=1))
boxes = A.upload(np.stack(frames).astype(np.float32))
n = frames[0].shape[0]
wb = L.raw("rd_nms3d_workspace_bytes")(n, B)
ws, keep, out = A.alloc(wb), A.alloc(B * mk * 4), A.alloc(B * mk * 40)
for thr in (0.2, 0.7):
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        L.call("rd_nms3d", A.ptr(boxes), B, n, thr, mk, 0, A.ptr(keep), A.ptr(out), A.ptr(ws), wb, A.stream)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    k = A.view_i32(keep, (B, mk)).cpu().numpy()
    print("N %d B %d max_keep %d thr %.1f: enqueue %.2f ms, total %.2f ms, kept %s, workspace %.1f MB" %
          (n, B, mk, thr, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (k >= 0).sum(axis=1).tolist(), wb / 1e6))
