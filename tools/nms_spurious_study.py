"""Characterisation of the REFERENCE's overlap routine on boxes that do not intersect, and validation of the weighted NMS's
rejection test against it (test infrastructure; needs `make -C oracle ref study`, i.e. the build container with /root/reference).

    python tools/nms_spurious_study.py FAMILY PAIRS_PER_JOB THREADS JOBS EPS_LO EPS_HI SEED0

Each job draws PAIRS_PER_JOB random pairs of a family (oracle/ref_overlap_study.cpp: 0 general position, 1 nearly parallel, 3 nearly
touching, 4 extreme sizes, 5/6/7/8 inside the rejection test's domain: every orientation / nearly parallel / nearly axis-parallel /
traffic-like), keeps the disjoint ones (separating-axis gap > 0 in double precision), runs nms.h:195-249 on each and counts the pairs
that the device's rejection test (restated there operation for operation) would skip although the reference's value is >= 1e-6.
profiles/r04_nms_spurious_study.txt is the log of the round-4 run (8.1e9 pairs, 0 violations)."""
import os, sys, time, numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref'))
import ref_overlap_study as S
fam, n_per, nthreads, reps = int(sys.argv[1]), int(float(sys.argv[2])), int(sys.argv[3]), int(sys.argv[4])
eps_lo, eps_hi = float(sys.argv[5]), float(sys.argv[6])
seed0 = int(sys.argv[7])
t0 = time.time()
tot = np.zeros(7, np.int64); rows = []
def job(i): return S.study_run(fam, seed0 + i, n_per, eps_lo, eps_hi, 0.0, False)
with ThreadPoolExecutor(nthreads) as ex:
    for r in ex.map(job, range(reps)):
        tot += np.array(r[1:], np.int64)
        if r[0].size: rows.append(np.array(r[0]).reshape(-1, 27))
rows = np.concatenate(rows) if rows else np.zeros((0, 27), np.float32)
print("family %d eps [%g, %g] seeds %d..%d: disjoint pairs %d, nan %d, negative %d, positive %d, >=0.05 %d | skippable %d (%.1f %%), VIOLATIONS (skippable and ovr >= 1e-6) %d | largest ovr among skippable %.3e | %.0fs" %
      (fam, eps_lo, eps_hi, seed0, seed0 + reps - 1, tot[0], tot[1], tot[2], tot[3], tot[4], tot[5], 100.0 * tot[5] / max(tot[0], 1), tot[6],
       rows[:, 24].max() if len(rows) else 0.0, time.time() - t0), flush=True)
if tot[6]: np.save('nms_study_violations_f%d_%d.npy' % (fam, seed0), rows)
