"""Round 3, batch u: per-tile cost of the two-problem form of the persistent conv: ONE plain launch over 2B frames against ONE pair
launch over B + B frames -- the same tile list, the same data statistics, the same bytes; what differs is the kernel form."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rangedet_amd import lib as R  # noqa: E402

L = R.get_lib()
H, DT = 64, R.RD_BF16
B = 8
st = torch.cuda.current_stream().cuda_stream
fl = R.RD_RELU_POST | R.RD_SCALE_FOLDED


def timeit(run, n=30):
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


for W in (2656, 1328, 664):
    cin = 128
    NB = 2
    xs = [torch.relu(torch.randn(2 * B * H * W * cin, device="cuda")).to(torch.bfloat16) for _ in range(NB)]
    ys = [torch.empty(2 * B * H * W * 128, device="cuda", dtype=torch.bfloat16) for _ in range(NB)]
    w = torch.from_numpy(L.pack_conv3x3_ex(np.random.randn(128, cin, 3, 3).astype(np.float32) * 0.05, 1, cin,
                                           fold_scale=np.ones(128, np.float32), dtype=DT)).cuda()
    sh = torch.zeros(128, device="cuda")
    half_x, half_y = B * H * W * cin * 2, B * H * W * 128 * 2   # bytes

    def single(i):
        L.call("rd_conv3x3_bn_act_ex", xs[i % NB].data_ptr(), cin, 0, w.data_ptr(), None, sh.data_ptr(), None, 0, 0, None, 0, 0, 0,
               None, ys[i % NB].data_ptr(), 128, 0, 2 * B, H, W, cin, 128, 1, fl, DT, st)

    def pair(i):
        x, y = xs[i % NB].data_ptr(), ys[i % NB].data_ptr()
        L.call("rd_conv3x3_bn_act_pair", x, 0, w.data_ptr(), sh.data_ptr(), y, 0,
               x + half_x, 0, w.data_ptr(), sh.data_ptr(), y + half_y, 0, cin, 128, B, H, W, cin, fl, DT, st)
    r = []
    for _ in range(3):
        r.append((timeit(single), timeit(pair)))
    a, b = min(v[0] for v in r), min(v[1] for v in r)
    print("W %-5d plain 2B launch %8.1f us   pair launch %8.1f us   %+5.2f %%" % (W, a, b, 100 * (b - a) / a), flush=True)
