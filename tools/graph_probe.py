"""Dev probe: does replaying the forward as a HIP graph beat stream launches?  python tools/graph_probe.py"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from rangedet_amd import lib as rdlib, synth  # noqa: E402
from rangedet_amd.pipeline import RangeDetPipeline  # noqa: E402

B = 8
params = synth.make_weights(seed=18)
fr = [synth.make_frame(i) for i in range(B)]
frame = {k: torch.from_numpy(np.concatenate([f[k] for f in fr])).cuda() for k in fr[0] if isinstance(fr[0][k], np.ndarray)}
pipe = RangeDetPipeline(params, dtype=rdlib.RD_BF16, batch=B)
for _ in range(3):
    pipe.forward(frame)
torch.cuda.synchronize()


def timeit(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


t_stream = timeit(lambda: pipe.forward(frame))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    pipe.forward(frame)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    pipe.forward(frame)
t_graph = timeit(g.replay)
print("graph-level forward of %d frames: stream launches %.3f ms, graph replay %.3f ms" % (B, t_stream, t_graph))
