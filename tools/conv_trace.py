"""Phase trace of the persistent 3x3 bf16 conv kernel (dev tool): python tools/conv_trace.py [W cin cout B]
Per workgroup: prologue, first tile's MFMA phase, first tile's epilogue, total life (s_memrealtime, 10 ns ticks)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rangedet_amd import lib as R  # noqa: E402

L = R.get_lib()
trace = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")       # the caller owns the trace buffer (the library never allocates)
L.cdll.rd_dev_conv_trace_set.argtypes = [ctypes.c_void_p, ctypes.c_long]
assert L.cdll.rd_dev_conv_trace_set(trace.data_ptr(), trace.numel()) == 0
W, cin, cout, B = [int(v) for v in (sys.argv[1:5] + ["2656", "128", "128", "8"][len(sys.argv) - 1:])]
H, dt = 64, R.RD_BF16
cs = -(-cin // 16) * 16
x = torch.randn(B * H * W * cs, device="cuda").to(torch.bfloat16)
y = torch.empty(B * H * W * cout, device="cuda", dtype=torch.bfloat16)
w = torch.from_numpy(L.pack_conv_weight(np.random.randn(cout, cin, 3, 3).astype(np.float32) * 0.05, dt)).cuda()
sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    L.call("rd_conv2d_bn_act", x.data_ptr(), cs, 0, w.data_ptr(), sc.data_ptr(), sh.data_ptr(), 0, cout, 0,
           y.data_ptr(), cout, 0, B, H, W, cin, cout, 3, 3, 1, 4, dt, st)
torch.cuda.synchronize()
ntiles = -(-W // 62) * -(-H // 8) * B
nwg = min(ntiles, torch.cuda.get_device_properties(0).multi_processor_count)
buf = np.zeros(nwg * 8, dtype=np.uint64)
fn = L.cdll.rd_dev_conv_trace_read
fn.argtypes = [ctypes.c_void_p, ctypes.c_long]
assert fn(buf.ctypes.data, buf.size) == 0
t = buf.reshape(nwg, 8).astype(np.int64)
clk = t[:, 7].copy(); t[:, 7] = 0
npt = int((t[0] > 0).sum())
t0 = t[:, 0].min()
print("tiles %d on %d workgroups (%.2f each), launch span %.1f us" % (ntiles, nwg, ntiles / nwg, (t[:, :npt].max() - t0) / 100.0))
names = ["prologue", "tile 0 MFMA phase", "tile 0 epilogue", "remaining tiles", "", ""]
for i in range(npt - 1):
    d = (t[:, i + 1] - t[:, i]) / 100.0
    print("  %-18s mean %8.2f us   p10 %8.2f  p90 %8.2f" % (names[i], d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
life = (t[:, npt - 1] - t[:, 0]) / 100.0
mhz = np.median(clk / life)
print("  shader clock %.0f MHz (s_memtime ticks / s_memrealtime)" % mhz)
print("  in shader cycles: MFMA phase %.1fk, epilogue %.1fk per tile" % ((t[:, 2] - t[:, 1]).mean() / 100.0 * mhz / 1e3,
                                                                          (t[:, 3] - t[:, 2]).mean() / 100.0 * mhz / 1e3))
print("  start skew p90 %.2f us" % np.percentile((t[:, 0] - t0) / 100.0, 90))
