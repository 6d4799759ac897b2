"""1 / 2 / 3 batches in flight on separate launch streams (dev experiment behind pipeline.InterleavedPipelines):
   python tools/two_stream.py [steps]     measured 13.5 / 12.5 / 12.9 ms per 8-frame step"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rangedet_amd import lib as rdlib, synth  # noqa: E402
from rangedet_amd.pipeline import RangeDetPipeline  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
P = synth.make_weights(seed=18)
pipes = [RangeDetPipeline(P, dtype=rdlib.RD_BF16, wnms_cap=4096, batch=8) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
fr = [synth.make_batch(list(range(8 * i, 8 * i + 8))) for i in range(2)]


def run(nstreams, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        j = i % nstreams
        with torch.cuda.stream(streams[j]):
            pipes[j].enqueue(fr[i % 2])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for ns in (1, 2, 3):
    run(ns, 6)
    ms = run(ns, steps)
    print("%d stream(s): %.2f ms per 8-frame step, %.1f frames/s" % (ns, ms, 8e3 / ms))
