"""Round 3, batch s: rd_conv3x3_bn_act_pair / rd_conv2d_bn_act_head_out_pair against the two single launches they replace
(head tower shapes, B frames, serial on one stream): us per pair."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rangedet_amd import lib as R  # noqa: E402

L = R.get_lib()
H, DT = 64, R.RD_BF16
B = int(os.environ.get("B", "8"))
st = torch.cuda.current_stream().cuda_stream
fl = R.RD_RELU_POST | R.RD_SCALE_FOLDED


def timeit(run, n=20):
    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for W, cin, head in ((2656, 128, False), (2656, 80, False), (2656, 128, True), (1328, 128, False), (1328, 128, True), (664, 128, False), (664, 128, True)):
    NB = 2
    xs = [[torch.relu(torch.randn(B * H * W * cin, device="cuda")).to(torch.bfloat16) for _ in range(2)] for _ in range(NB)]
    ys = [[torch.empty(B * H * W * 128, device="cuda", dtype=torch.bfloat16) for _ in range(2)] for _ in range(NB)]
    ws = [torch.from_numpy(L.pack_conv3x3_ex(np.random.randn(128, cin, 3, 3).astype(np.float32) * 0.05, 1, cin,
                                             fold_scale=np.ones(128, np.float32), dtype=DT)).cuda() for _ in range(2)]
    sh = [torch.zeros(128, device="cuda") for _ in range(2)]
    nouts = (1, 8)
    hw = [torch.from_numpy(L.pack_head_weight(np.random.randn(n, 128).astype(np.float32) * 0.1, dtype=DT)).cuda() for n in nouts]
    hb = [torch.zeros(n, device="cuda") for n in nouts]
    ho = [torch.empty(B * H * W * n, device="cuda") for n in nouts]

    def single(i):
        for g in range(2):
            if head:
                L.call("rd_conv2d_bn_act_head_out", xs[i % NB][g].data_ptr(), cin, 0, ws[g].data_ptr(), None, sh[g].data_ptr(), B, H, W, cin, fl,
                       hw[g].data_ptr(), hb[g].data_ptr(), ho[g].data_ptr(), H * W * nouts[g], 0, nouts[g], DT, st)
            else:
                L.call("rd_conv3x3_bn_act_ex", xs[i % NB][g].data_ptr(), cin, 0, ws[g].data_ptr(), None, sh[g].data_ptr(), None, 0, 0, None, 0, 0, 0,
                       None, ys[i % NB][g].data_ptr(), 128, 0, B, H, W, cin, 128, 1, fl, DT, st)

    def pair(i):
        x = xs[i % NB]
        if head:
            L.call("rd_conv2d_bn_act_head_out_pair",
                   x[0].data_ptr(), 0, ws[0].data_ptr(), sh[0].data_ptr(), hw[0].data_ptr(), hb[0].data_ptr(), ho[0].data_ptr(), H * W * nouts[0], nouts[0],
                   x[1].data_ptr(), 0, ws[1].data_ptr(), sh[1].data_ptr(), hw[1].data_ptr(), hb[1].data_ptr(), ho[1].data_ptr(), H * W * nouts[1], nouts[1],
                   cin, 0, B, H, W, cin, fl, DT, st)
        else:
            y = ys[i % NB]
            L.call("rd_conv3x3_bn_act_pair", x[0].data_ptr(), 0, ws[0].data_ptr(), sh[0].data_ptr(), y[0].data_ptr(), 0,
                   x[1].data_ptr(), 0, ws[1].data_ptr(), sh[1].data_ptr(), y[1].data_ptr(), 0, cin, 128, B, H, W, cin, fl, DT, st)
    a, b = timeit(single), timeit(pair)
    print("W %-5d cin %-3d %-9s two launches %8.1f us   one pair launch %8.1f us   %+5.1f %%" % (W, cin, "+out conv" if head else "", a, b, 100 * (b - a) / a), flush=True)
