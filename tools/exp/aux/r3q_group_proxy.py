"""Round 3, batch q: what would ONE launch for two equal-shaped convs buy (cls + reg tower layers; a cout-128 layer as two cout-64
halves)?  Proxy through the production entry point: the same conv at 2B frames in one launch (= the tile list of a two-group launch)
against two launches at B frames, serial on one stream.  Prints us per PAIR of B-frame problems."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rangedet_amd import lib as R  # noqa: E402

L = R.get_lib()
H, DT = 64, R.RD_BF16
st = torch.cuda.current_stream().cuda_stream


def time_conv(W, cin, cout, B, nlaunch, res=False, n=20):
    NB = 3
    xs = [torch.randn(B * H * W * cin, device="cuda").to(torch.bfloat16) for _ in range(NB)]
    ys = [torch.empty(B * H * W * cout, device="cuda", dtype=torch.bfloat16) for _ in range(NB)]
    rs = [torch.randn(B * H * W * cout, device="cuda").to(torch.bfloat16) for _ in range(NB)]
    w = torch.from_numpy(L.pack_conv3x3_ex(np.random.randn(cout, cin, 3, 3).astype(np.float32) * 0.05, 1, cin,
                                           fold_scale=np.ones(cout, np.float32), dtype=DT)).cuda()
    sh = torch.zeros(cout, device="cuda")
    fl = R.RD_RELU_POST | R.RD_SCALE_FOLDED | (R.RD_ADD if res else 0)

    def run(i):
        L.call("rd_conv3x3_bn_act_ex", xs[i % NB].data_ptr(), cin, 0, w.data_ptr(), None, sh.data_ptr(),
               rs[i % NB].data_ptr() if res else None, cout if res else 0, 0, None, 0, 0, 0, None,
               ys[i % NB].data_ptr(), cout, 0, B, H, W, cin, cout, 1, fl, DT, st)
    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n * nlaunch):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


B = int(os.environ.get("B", "8"))
print("two B=%d launches vs one 2B launch (same tiles), 128->128, us per pair" % B)
for W in (2656, 1328, 664, 332, 166):
    a = time_conv(W, 128, 128, B, 2)
    b = time_conv(W, 128, 128, 2 * B, 1)
    print("  W %-5d  2 x B: %8.1f   1 x 2B: %8.1f   %+5.1f %%" % (W, a, b, 100 * (b - a) / a), flush=True)
print("72->128 (level-0 tower layer 0)")
a = time_conv(2656, 80, 128, B, 2)
b = time_conv(2656, 80, 128, 2 * B, 1)
print("  W 2656   2 x B: %8.1f   1 x 2B: %8.1f   %+5.1f %%" % (a, b, 100 * (b - a) / a), flush=True)
print("cout split: 128->128 at B vs 128->64 at 2B (two cout-64 halves per tile), us per layer")
for W in (664, 332, 166):
    for res in (False, True):
        a = time_conv(W, 128, 128, B, 1, res)
        b = time_conv(W, 128, 64, 2 * B, 1, res)
        print("  W %-5d %-4s 128->128: %8.1f   2 x (128->64): %8.1f   %+5.1f %%" % (W, "+add" if res else "", a, b, 100 * (b - a) / a), flush=True)
