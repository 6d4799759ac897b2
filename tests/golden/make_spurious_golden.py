"""Generates tests/golden/pair_overlaps_spurious.npz and tests/golden/wnms_k90_spurious.npz: pairs of boxes that do NOT intersect
for which the REFERENCE's overlap routine (nms.h:195-249, compiled as-is: `make -C oracle ref study`) nevertheless returns a
positive value -- edge directions that tie within its EPS, nearly parallel boxes, ill-conditioned geometry -- and a weighted-NMS
case built from such pairs, in which the reference suppresses boxes that touch nothing.  Data only: inputs and the reference's
outputs.  Run in the build container:   python tests/golden/make_spurious_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
import processing_cxx_ref as ref  # noqa: E402
import ref_overlap_study as S  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def tie_margin(a, b):
    """Distance of the pair from the clipper's angle tie |angle_i - angle_j| < 1e-5 (nms.h:58-64,104-106): the smallest
    | |difference| - 1e-5 | over the 16 edge pairs in both orientations (a reversed winding shifts an angle by pi).  The device's
    atan2f differs from glibc's by an ulp or two (~3e-7 near pi), so a pair with a margin of a few 1e-6 gets the same ties on
    both; a pair closer to the threshold may legitimately fall on either side."""
    pa, pb = a[:, :8].reshape(-1, 4, 2), b[:, :8].reshape(-1, 4, 2)
    va, vb = np.roll(pa, -1, 1) - pa, np.roll(pb, -1, 1) - pb
    best = np.full(len(a), np.inf)
    for sa in (1, -1):
        for sb in (1, -1):
            ga = np.arctan2((sa * va[..., 1]).astype(np.float32), (sa * va[..., 0]).astype(np.float32)).astype(np.float64)
            gb = np.arctan2((sb * vb[..., 1]).astype(np.float32), (sb * vb[..., 0]).astype(np.float32)).astype(np.float64)
            d = np.abs(ga[:, :, None] - gb[:, None, :])
            d = np.minimum(d, np.abs(d - 2 * np.pi))
            best = np.minimum(best, np.abs(d - 1e-5).min((1, 2)))
    return best


rng = np.random.default_rng(4)
# (family, pairs drawn, eps_lo, eps_hi): see oracle/ref_overlap_study.cpp
FAMILIES = [(0, 4e7, 0, 0), (1, 2e5, 1e-8, 1e-2), (2, 2e5, 0, 0), (3, 2e5, 0, 0), (4, 4e6, 0, 0), (5, 4e7, 0, 0), (6, 4e6, 2e-5, 5e-2),
            (7, 4e6, 1e-8, 1e-1), (8, 4e7, 0, 0)]
A, B, O, F = [], [], [], []
for fam, n, lo, hi in FAMILIES:
    out = np.array(S.study_run(fam, 7 + fam, int(n), lo, hi, 0.0, True)[0]).reshape(-1, 27)
    o = out[:, 24]
    big, small = np.nonzero(o >= 1e-3)[0], np.nonzero(o < 1e-3)[0]
    sel = np.concatenate([rng.permutation(big)[:300], rng.permutation(small)[:60]])
    print("family %d: %d positive results on disjoint pairs (%d >= 1e-3), kept %d" % (fam, len(o), len(big), len(sel)))
    A.append(out[sel, :12]); B.append(out[sel, 12:24]); O.append(o[sel]); F.append(np.full(len(sel), fam, np.int32))
a, b, o, f = (np.concatenate(v) for v in (A, B, O, F))
chk = np.array(ref.pair_overlaps(a, b, False), np.float32)
assert np.array_equal(chk, o)
np.savez_compressed(os.path.join(HERE, "pair_overlaps_spurious.npz"), a=a, b=b, iou=o, iou3d=np.array(ref.pair_overlaps(a, b, True), np.float32), family=f,
                    tie_margin=tie_margin(a, b).astype(np.float32))



# a weighted-NMS case: 120 pairs whose first box "overlaps" its disjoint partner by >= 0.1 for the reference; the first boxes get the
# higher scores (they are the kept box i of nms.h:502-517), every score is distinct
# (only pairs at least 3e-6 rad away from the angle-tie threshold: the same result with glibc's and the device's atan2f)
m = (o >= 0.1) & (np.abs(a[:, :8]).max(1) < 95) & (np.abs(b[:, :8]).max(1) < 95) & (tie_margin(a, b) > 3e-6)
idx = rng.permutation(np.nonzero(m)[0])[:120]
d = np.concatenate([a[idx], b[idx]]).astype(np.float32)
d[:, 11] = np.concatenate([0.95 - 1e-3 * np.arange(len(idx)), 0.60 - 1e-3 * np.arange(len(idx))]).astype(np.float32)
rows, keep = ref.wnms_4c(d, 0.1, 0.5, False, 100)
print("wnms_k90_spurious: %d boxes, %d kept (the reference suppresses %d boxes that intersect nothing)" % (len(d), len(keep), len(d) - len(keep)))
np.savez_compressed(os.path.join(HERE, "wnms_k90_spurious.npz"), dets=d, thresh=0.1, thresh_vote=0.5, is3d=False, hash_scale=100,
                    rows=np.array(rows, np.float32).reshape(-1, 12), keep=np.array(keep, np.int32))
