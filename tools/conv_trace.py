"""Phase trace of the bf16 conv kernel (dev tool): RD_CONV_TRACE=1 python tools/conv_trace.py [W cin cout B]
Prints, per phase, the mean / p90 duration over workgroups and the launch-relative start spread (s_memrealtime, 10 ns)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RD_CONV_TRACE", "1")
from rangedet_amd import lib as R  # noqa: E402

L = R.get_lib()
W, cin, cout, B = [int(v) for v in (sys.argv[1:5] + ["2656", "128", "128", "8"][len(sys.argv) - 1:])]
H, dt = 64, R.RD_BF16
cs = -(-cin // 16) * 16
x = torch.randn(B * H * W * cs, device="cuda").to(torch.bfloat16)
y = torch.empty(B * H * W * cout, device="cuda", dtype=torch.bfloat16)
w = torch.from_numpy(L.pack_conv_weight(np.random.randn(cout, cin, 3, 3).astype(np.float32) * 0.05, dt)).cuda()
sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    L.call("rd_conv2d_bn_act", x.data_ptr(), cs, 0, w.data_ptr(), sc.data_ptr(), sh.data_ptr(), 0, cout, 0,
           y.data_ptr(), cout, 0, B, H, W, cin, cout, 3, 3, 1, 4, dt, st)
torch.cuda.synchronize()
nwg = ((W + 63) // 64) * (H // 4) * B * (1 if os.environ.get("RD_CONV_V2") else cout // 64)
buf = np.zeros(nwg * 8, dtype=np.uint64)
fn = L.cdll.rd_dev_conv_trace_read
fn.argtypes = [ctypes.c_void_p, ctypes.c_long]
assert fn(buf.ctypes.data, buf.size) == 0
t = buf.reshape(nwg, 8).astype(np.int64)
npt = int((t[0] > 0).sum())
t0 = t[:, 0].min()
print("workgroups %d, trace points %d, launch span %.1f us" % (nwg, npt, (t[:, :npt].max() - t0) / 100.0))
names = ["prologue+stage0", "chunk0 taps", "stage1", "chunk1 taps", "epilogue", "", ""]
if not os.environ.get("RD_CONV_V2"):
    names = ["prologue+stage0", "chunk0 taps", "barrier+stage1", "chunk1 taps", "epilogue", "", ""]
for i in range(npt - 1):
    d = (t[:, i + 1] - t[:, i]) / 100.0
    print("  %-16s mean %7.2f us   p10 %7.2f  p90 %7.2f" % (names[i], d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
d = (t[:, npt - 1] - t[:, 0]) / 100.0
print("  %-16s mean %7.2f us   p10 %7.2f  p90 %7.2f" % ("workgroup life", d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
