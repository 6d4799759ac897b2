"""How full is the GPU in the timed pipeline?  From a rocprofv3 --kernel-trace CSV of `bench.py` (three batches in flight): over the middle of
the trace, the share of wall time with 0 / 1 / 2 / 3+ kernels running, the share with at least one MFMA kernel (conv3x3_stream / block64_stream /
meta16) running, the time-weighted sum of resident-slot demand (min(grid, 512) workgroups of the persistent kernels), and how much longer the
persistent launches take here than alone (per kernel name: mean duration in this trace).
    python tools/overlap_trace.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    wg = max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256)) or 256))
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // wg
    ev.append((s, e, name, grid))
ev.sort()
t0, t1 = ev[0][0], max(e for _, e, _, _ in ev)
lo, hi = t0 + (t1 - t0) * 0.3, t0 + (t1 - t0) * 0.9          # the timed regions (warm-up and set-up are in the first part)
pts = []
for s, e, name, grid in ev:
    if e <= lo or s >= hi:
        continue
    mf = any(k in name for k in ("conv3x3_stream", "block64_stream", "meta16"))
    pts.append((max(s, lo), 1, mf, min(grid, 512) if mf else 0))
    pts.append((min(e, hi), -1, mf, min(grid, 512) if mf else 0))
pts.sort(key=lambda p: (p[0], p[1]))
n = nm = slots = 0
last = lo
hist, hist_m = defaultdict(float), defaultdict(float)
slot_t = 0.0
for t, d, mf, g in pts:
    dt = t - last
    hist[min(n, 4)] += dt
    hist_m[min(nm, 4)] += dt
    slot_t += dt * min(slots, 1024)
    last = t
    n += d
    if mf:
        nm += d
        slots += d * g
W = hi - lo
print("window %.1f ms, %d kernel launches in it" % (W / 1e6, len(pts) // 2))
print("kernels running   0: %5.1f %%  1: %5.1f %%  2: %5.1f %%  3: %5.1f %%  4+: %5.1f %%" % tuple(100 * hist[k] / W for k in range(5)))
print("MFMA kernels      0: %5.1f %%  1: %5.1f %%  2: %5.1f %%  3: %5.1f %%  4+: %5.1f %%" % tuple(100 * hist_m[k] / W for k in range(5)))
print("time-weighted workgroup demand of the running MFMA kernels: %.0f of 512 resident slots" % (slot_t / W))
dur = defaultdict(list)
for s, e, name, grid in ev:
    if s >= lo and e <= hi:
        dur[name.split("(")[0][-70:]].append((e - s) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("%-72s %5d launches  mean %8.1f us  sum %8.1f ms" % (k, len(v), sum(v) / len(v), sum(v) / 1e3))
