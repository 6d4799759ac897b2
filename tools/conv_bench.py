"""Conv micro-benchmark on the GPU (tuning aid): selected layer shapes of the RangeDet graph, us + TFLOP/s.
   [B=<frames>] [ONLY=<substring>] python tools/conv_bench.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rangedet_amd import lib as R  # noqa: E402

L = R.get_lib()
dt = R.RD_BF16
H = 64
B = int(os.environ.get("B", "8"))     # frames per launch (bench.py default batch)
CASES = [  # name, W, cin, cout, k, stride, flags
    ("head_l0 128->128", 2656, 128, 128, 3, 1, 4), ("head_l0 72->128", 2656, 72, 128, 3, 1, 4),
    ("res1 64->64", 2656, 64, 64, 3, 1, 4), ("res1 64->64 +add", 2656, 64, 64, 3, 1, 6),
    ("head_l1 128->128", 1328, 128, 128, 3, 1, 4), ("res2 128->128 W664", 664, 128, 128, 3, 1, 4),
    ("res3a 128->128 W332", 332, 128, 128, 3, 1, 4), ("res3 128->128 W166", 166, 128, 128, 3, 1, 4),
    ("res3 +add W166", 166, 128, 128, 3, 1, 6), ("sc 64->128 s2", 1328, 64, 128, 1, 2, 0), ("conv 8->64", 2656, 8, 64, 3, 1, 4),
    ("sc 64->64 s1", 2656, 64, 64, 1, 1, 0), ("sc 8->64 s1", 2656, 8, 64, 1, 1, 0), ("sc 64->64 s2", 2656, 64, 64, 1, 2, 0),
    ("sc 128->128 s2 W664", 664, 128, 128, 1, 2, 0), ("sc 128->128 s1 W664", 664, 128, 128, 1, 1, 0),
]
st = torch.cuda.current_stream().cuda_stream
ONLY = os.environ.get("ONLY")
for name, W, cin, cout, k, s, fl in CASES:
    if ONLY and ONLY not in name:
        continue
    cs = -(-cin // 16) * 16
    Wout = (W + 2 * (k // 2) - k) // s + 1
    x = torch.randn(B * H * W * cs, device="cuda").to(torch.bfloat16)
    y = torch.empty(B * H * Wout * cout, device="cuda", dtype=torch.bfloat16)
    r = torch.randn(B * H * Wout * cout, device="cuda").to(torch.bfloat16)
    w = torch.from_numpy(L.pack_conv_weight(np.random.randn(cout, cin, k, k).astype(np.float32) * 0.05, dt)).cuda()
    sc = torch.ones(cout, device="cuda")
    sh = torch.zeros(cout, device="cuda")

    def run():
        L.call("rd_conv2d_bn_act", x.data_ptr(), cs, 0, w.data_ptr(), sc.data_ptr(), sh.data_ptr(), r.data_ptr(), cout, 0,
               y.data_ptr(), cout, 0, B, H, W, cin, cout, k, k, s, fl, dt, st)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    flops = 2.0 * B * H * Wout * cin * cout * k * k
    print("%-22s %8.1f us %8.1f TFLOP/s" % (name, us, flops / us / 1e6), flush=True)
