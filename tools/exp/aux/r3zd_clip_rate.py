"""Round 3, batch zd: what the weighted-NMS pair kernel has to do on the bench's own candidates, and what a clip costs.
Counts, per frame, the pairs of the two rounds that share a BBoxHash cell; times rd_single_overlap on exactly those pairs
(one pair per lane, dense) = the polygon clips alone."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rangedet_amd import lib as rdlib, synth  # noqa: E402
from rangedet_amd.pipeline import RangeDetPipeline  # noqa: E402

pipe = RangeDetPipeline(synth.make_weights(seed=18), dtype=rdlib.RD_BF16, wnms_cap=4096, batch=8)
fr = synth.make_batch(list(range(8)))
pipe.enqueue(fr)
torch.cuda.synchronize()
bp = pipe.bpost
A, L = bp.A, bp.L
cnt = A.view_i32(bp.count, (8,)).cpu().numpy()
dets = A.view_f32(bp.dets, (8, bp.k, 12)).cpu().numpy()
tot1 = tot_all = 0
pa, pb = [], []
for b in range(8):
    K = int(cnt[b])
    d = dets[b, :K]
    xs, ys = d[:, 0:8:2], d[:, 1:8:2]
    c0, c1 = np.floor(xs.min(1) / 100).astype(int), np.floor(ys.min(1) / 100).astype(int)
    c2, c3 = np.ceil(np.maximum(xs.max(1), 1.2e-38) / 100).astype(int), np.ceil(np.maximum(ys.max(1), 1.2e-38) / 100).astype(int)
    share = (np.maximum(c0[:, None], c0[None]) < np.minimum(c2[:, None], c2[None])) & (np.maximum(c1[:, None], c1[None]) < np.minimum(c3[:, None], c3[None]))
    iu = np.triu(np.ones((K, K), bool), 1)
    n_all = int((share & iu).sum())
    n1 = int((share & iu)[:256].sum())
    tot_all += n_all
    tot1 += n1
    if b == 0:
        i, j = np.nonzero((share & iu)[:256])
        pa, pb = d[i], d[j]
        yaw = d[:, 8]
        print("frame 0: K %d, pairs sharing a cell: all %d of %d (%.2f), round-1 rows %d; yaw spread %.3f; cells x [%d, %d) y [%d, %d)" % (
            K, n_all, K * (K - 1) // 2, n_all / (K * (K - 1) / 2), n1, float(yaw.std()), c0.min(), c2.max(), c1.min(), c3.max()))
print("8 frames: %d clips in round 1, %d if every row were evaluated" % (tot1, tot_all))
n = len(pa)
da, db = torch.from_numpy(np.ascontiguousarray(pa)).cuda(), torch.from_numpy(np.ascontiguousarray(pb)).cuda()
out = torch.empty(n, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    L.call("rd_single_overlap", da.data_ptr(), db.data_ptr(), n, 0, out.data_ptr(), st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    L.call("rd_single_overlap", da.data_ptr(), db.data_ptr(), n, 0, out.data_ptr(), st)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
o = out.cpu().numpy()
print("rd_single_overlap on frame 0's %d round-1 pairs: %.1f us = %.2f G clips/s; %d with overlap > 0, %d >= 0.5" % (n, us, n / us / 1e3, int((o > 0).sum()), int((o >= 0.5).sum())))
