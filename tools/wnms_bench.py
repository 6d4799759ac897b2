"""Batched post-processing on the bench's own candidates (dev tool): python tools/wnms_bench.py
   RD_WNMS_ONE_ROUND=1 switches the weighted NMS back to a single pairs/scan round.
(Round 6: the RD_* variables named here are DEVELOPMENT switches -- the release library ignores them.  Build the A/B library with
`python -m rangedet_amd.build --dev` and run with RANGEDET_HIP_LIB=rangedet_amd/librangedet_hip_dev.so RD_DEV_SWITCHES=1; tools/exp/ab.sh does both.)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rangedet_amd import lib as rdlib, synth  # noqa: E402
from rangedet_amd.pipeline import RangeDetPipeline  # noqa: E402

pipe = RangeDetPipeline(synth.make_weights(seed=18), dtype=rdlib.RD_BF16, wnms_cap=4096, batch=8)
fr = synth.make_batch(list(range(8)))
pipe.enqueue(fr)
torch.cuda.synchronize()
res = [p.collect() for p in pipe.post]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    pipe.bpost.enqueue_nms()
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    pipe.bpost.enqueue_nms()
e1.record()
torch.cuda.synchronize()
print("batched NMS of 8 frames (%s candidates, %s kept): %.1f us per batch" % (
    [r["num_candidates"] for r in res], [len(r["keep_inds"]) for r in res], e0.elapsed_time(e1) * 1e3 / 20))
