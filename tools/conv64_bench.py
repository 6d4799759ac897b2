"""The HBM-bound 64->64 3x3 layers through the PRODUCTION entry point (rd_conv3x3_bn_act_ex, folded scales, 8 x 30 tiles): us,
TFLOP/s and TB/s of algorithmic bytes, with and without the residual.  Tuning aid; dev switches (RD_CONV_HB3, and RD_CONV3_DBG
with a -DRD_CONV3_DEV build selected through RANGEDET_HIP_LIB) are read by the library.
    [B=8] [WS=2656,1328] [C128=1] [RES=0|1|both] [ITER=30] python tools/conv64_bench.py
(RES / ITER / WS select ONE launch class for a counter pass: tools/pmc_conv_classes.sh)
(Round 6: the RD_* variables named here are DEVELOPMENT switches -- the release library ignores them.  Build the A/B library with
`python -m rangedet_amd.build --dev` and run with RANGEDET_HIP_LIB=rangedet_amd/librangedet_hip_dev.so RD_DEV_SWITCHES=1; tools/exp/ab.sh does both.)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rangedet_amd import lib as R  # noqa: E402

L = R.get_lib()
H = 64
B = int(os.environ.get("B", "8"))
st = torch.cuda.current_stream().cuda_stream
DT = R.RD_F16 if os.environ.get("F16") else R.RD_BF16
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in ("RD_CONV_HB3", "RD_CONV3_DBG") if k in os.environ)
for W in [int(v) for v in os.environ.get("WS", "2656,1328").split(",")]:
    for cin, cout in {"1": ((64, 64), (128, 128)), "only": ((128, 128),)}.get(os.environ.get("C128", ""), ((64, 64),)):
        # several distinct buffers cycled so that no launch finds its input in the Infinity Cache by accident of the benchmark
        NB = 3
        xs = [torch.randn(B * H * W * cin, device="cuda").to(torch.float16 if DT == R.RD_F16 else torch.bfloat16) for _ in range(NB)]
        ys = [torch.empty(B * H * W * cout, device="cuda", dtype=torch.float16 if DT == R.RD_F16 else torch.bfloat16) for _ in range(NB)]
        rs = [torch.randn(B * H * W * cout, device="cuda").to(torch.float16 if DT == R.RD_F16 else torch.bfloat16) for _ in range(NB)]
        w = torch.from_numpy(L.pack_conv3x3_ex(np.random.randn(cout, cin, 3, 3).astype(np.float32) * 0.05, 1, cin,
                                               fold_scale=np.ones(cout, np.float32), dtype=DT)).cuda()
        sh = torch.zeros(cout, device="cuda")
        for res in {"0": (False,), "1": (True,)}.get(os.environ.get("RES", "both"), (False, True)):
            fl = R.RD_RELU_POST | R.RD_SCALE_FOLDED | (R.RD_ADD if res else 0)

            def run(i):
                L.call("rd_conv3x3_bn_act_ex", xs[i % NB].data_ptr(), cin, 0, w.data_ptr(), None, sh.data_ptr(),
                       rs[i % NB].data_ptr() if res else None, cout if res else 0, 0, None, 0, 0, 0, None,
                       ys[i % NB].data_ptr(), cout, 0, B, H, W, cin, cout, 1, fl, DT, st)
            for i in range(3):
                run(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = int(os.environ.get("ITER", "30"))
            e0.record()
            for i in range(n):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            flops = 2.0 * B * H * W * cin * cout * 9
            by = B * H * W * 2.0 * (cin + cout + (cout if res else 0))
            print("%-28s %d->%d W%-5d %-4s %8.1f us %7.1f TFLOP/s %5.2f TB/s" % (tag, cin, cout, W, "+add" if res else "", us,
                                                                                 flops / us / 1e6, by / us / 1e6), flush=True)
