"""Shader clock and launch time of the PRODUCTION 3x3 conv forms (rd_conv3x3_bn_act_ex, folded scales, 8 x 30 tiles) against the
statistics of the input: random bf16, post-ReLU random (half zeros, like real activations), all zero.  The in-kernel trace
(rd_dev_conv_trace_set) stamps s_memrealtime (100 MHz) at the start / end of every workgroup and its life in shader cycles (s_memtime).
    python tools/conv_clock.py [W cin cout B]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from rangedet_amd import lib as R  # noqa: E402
from power_sample import PowerSampler  # noqa: E402

L = R.get_lib()
trace = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")       # the caller owns the trace buffer (the library never allocates)
L.cdll.rd_dev_conv_trace_set.argtypes = [ctypes.c_void_p, ctypes.c_long]
assert L.cdll.rd_dev_conv_trace_set(trace.data_ptr(), trace.numel()) == 0
W, cin, cout, B = [int(v) for v in (sys.argv[1:5] + ["2656", "128", "128", "8"][len(sys.argv) - 1:])]
H = 64
st = torch.cuda.current_stream().cuda_stream
w = torch.from_numpy(L.pack_conv3x3_ex(np.random.randn(cout, cin, 3, 3).astype(np.float32) * 0.05, 1, cin,
                                       fold_scale=np.ones(cout, np.float32))).cuda()
sh = torch.zeros(cout, device="cuda")
y = torch.empty(B * H * W * cout, device="cuda", dtype=torch.bfloat16)
fn = L.cdll.rd_dev_conv_trace_read
fn.argtypes = [ctypes.c_void_p, ctypes.c_long]
nwg = 2 * torch.cuda.get_device_properties(0).multi_processor_count
for name, gen in (("random", lambda n: torch.randn(n, device="cuda")), ("post-ReLU random", lambda n: torch.relu(torch.randn(n, device="cuda"))),
                  ("all zero", lambda n: torch.zeros(n, device="cuda"))):
    x = gen(B * H * W * cin).to(torch.bfloat16)

    def run():
        L.call("rd_conv3x3_bn_act_ex", x.data_ptr(), cin, 0, w.data_ptr(), None, sh.data_ptr(), None, 0, 0, None, 0, 0, 0, None,
               y.data_ptr(), cout, 0, B, H, W, cin, cout, 1, R.RD_RELU_POST | R.RD_SCALE_FOLDED, R.RD_BF16, st)
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with PowerSampler() as ps:
        for _ in range(200):                                      # ~60 ms under the sampler first: the power management settles in ms
            run()
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
    buf = np.zeros(nwg * 8, dtype=np.uint64)
    assert fn(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(nwg, 8).astype(np.int64)
    clk = t[:, 7].copy()
    t[:, 7] = 0
    npt = int((t[0] > 0).sum())
    life = (t[:, npt - 1] - t[:, 0]) / 100.0                      # us
    print("%d->%d W%d B%d  %-18s %7.1f us/launch   shader clock %4.0f MHz (median over %d workgroups)" %
          (cin, cout, W, B, name, e0.elapsed_time(e1) * 1e3 / 20, np.median(clk / life), nwg), ps.summary(), flush=True)
