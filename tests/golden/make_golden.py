"""Generates tests/golden/*.npz from the REFERENCE's own wnms (oracle/_ref, built from /root/reference by
`make -C oracle ref`).  Data only: inputs and the reference's outputs.  Run in the build container:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cpu_ops as O  # noqa: E402
from rangedet_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ref = O.ref_module()
assert ref is not None, "build oracle/_ref first (make -C oracle ref)"


def save(tag, d, thr=0.1, vote=0.5, is3d=False, hs=100):
    rows, keep = ref.wnms_4c(d, thr, vote, is3d, hs)
    np.savez_compressed(os.path.join(HERE, "wnms_%s.npz" % tag), dets=d, thresh=thr, thresh_vote=vote, is3d=is3d,
                        hash_scale=hs, rows=np.array(rows, np.float32).reshape(-1, 12), keep=np.array(keep, np.int32))


save("k0", np.zeros((0, 12), np.float32))
save("k1", synth.cluster_dets(1, 1, seed=1))
save("k2", synth.cluster_dets(1, 2, seed=2))
save("k3", synth.cluster_dets(1, 3, seed=3))
save("k64", synth.cluster_dets(8, 8, seed=4))
save("k512", synth.cluster_dets(32, 16, seed=5))
save("k2048", synth.cluster_dets(128, 16, seed=6))
save("k512_3d", synth.cluster_dets(32, 16, seed=7), is3d=True)
save("k512_ties", synth.cluster_dets(32, 16, seed=8, quant=33))
save("k256_far", synth.cluster_dets(16, 16, seed=9, spread=170.0), hs=100)
save("k256_hash10", synth.cluster_dets(16, 16, seed=10), hs=10)
save("k256_thr", synth.cluster_dets(16, 16, seed=11, jitter=0.6), thr=0.3, vote=0.6)
d = synth.cluster_dets(64, 16, seed=12)
rng = np.random.default_rng(0)
a = d[rng.integers(0, len(d), 4000)]
b = d[rng.integers(0, len(d), 4000)]
near = np.abs(a[:, :2] - b[:, :2]).sum(1) < 8  # keep mostly overlapping pairs + a sample of the rest
sel = near | (rng.uniform(size=4000) < 0.1)
a, b = a[sel], b[sel]
np.savez_compressed(os.path.join(HERE, "pair_overlaps.npz"), a=a, b=b,
                    iou=np.array(ref.pair_overlaps(a, b, False), np.float32),
                    iou3d=np.array(ref.pair_overlaps(a, b, True), np.float32))
print("golden vectors written to", HERE)
