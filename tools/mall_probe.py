"""Does the producer -> consumer tensor of a conv chain come back from the 256-MB memory-side cache when it is small enough?
A tower-like chain of three 128 -> 128 convs (16 x 16 x 32 form) over 8 frames of 64 x 2048 (512 tiles per frame: every sub-batch size is a
whole number of tile rounds, so launch quantisation does not enter), run depth-first over sub-batches of 8 / 4 / 2 / 1 frames: the tensor
between two convs is 268 / 134 / 67 / 34 MB.  Prints us per 8 frames x 3 convs, board power and shader clock for each sub-batch size.
    python tools/mall_probe.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from rangedet_amd import lib as R  # noqa: E402
from power_sample import PowerSampler  # noqa: E402

L = R.get_lib()
st = torch.cuda.current_stream().cuda_stream
B, H, dt = 8, 64, R.RD_BF16
rng = np.random.default_rng(0)
NCONV = 3


def run(W, C, sub, secs=2.0):
    w = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
    fs = rng.uniform(0.5, 1.5, C).astype(np.float32)
    sh = torch.zeros(C, device="cuda")
    m16 = C == 128
    wp = torch.from_numpy(L.pack_conv3x3_m16(w, fs, dt) if m16 else L.pack_conv3x3_ex(w, 1, C, fold_scale=fs, dtype=dt)).cuda()
    fl = R.RD_RELU_POST | R.RD_SCALE_FOLDED | (R.RD_MFMA16 if m16 else 0)
    bufs = [torch.relu(torch.randn(B, H, W, C, device="cuda")).to(torch.bfloat16) for _ in range(NCONV + 1)]
    fb = H * W * C * 2   # bytes per frame

    def chain():
        for s in range(0, B, sub):
            for i in range(NCONV):
                L.call("rd_conv3x3_bn_act_ex", bufs[i].data_ptr() + s * fb, C, 0, wp.data_ptr(), None, sh.data_ptr(), None, 0, 0, None, 0, 0, 0, None,
                       bufs[i + 1].data_ptr() + s * fb, C, 0, sub, H, W, C, C, 1, fl, dt, st)

    for _ in range(5):
        chain()
    torch.cuda.synchronize()
    with PowerSampler(period=0.01) as ps:
        t0 = time.time()
        n = 0
        while time.time() - t0 < secs:
            for _ in range(10):
                chain()
            torch.cuda.synchronize()
            n += 10
        dtm = time.time() - t0
    fl_total = 2.0 * B * H * W * C * C * 9 * NCONV
    print("C %3d W %4d sub-batch %d (%5.1f MB between convs) %8.1f us per 8 frames x %d convs  %6.0f TFLOP/s   %s"
          % (C, W, sub, sub * fb / 1e6, dtm / n * 1e6, NCONV, fl_total * n / dtm / 1e12, ps.summary()), flush=True)


for rep in range(2):
    for W, C in ((2048, 128), (2048, 64)):
        for sub in (8, 4, 2, 1):
            run(W, C, sub)
