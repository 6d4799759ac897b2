"""Run ONE conv shape a few times (for rocprofv3 --pmc).  python tools/conv_one.py W cin cout k stride flags [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rangedet_amd import lib as R  # noqa: E402

W, cin, cout, k, s, fl = [int(v) for v in sys.argv[1:7]]
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
L = R.get_lib()
dt, H = R.RD_BF16, 64
cs = -(-cin // 16) * 16
Wout = (W + 2 * (k // 2) - k) // s + 1
x = torch.randn(H * W * cs, device="cuda").to(torch.bfloat16)
y = torch.empty(H * Wout * cout, device="cuda", dtype=torch.bfloat16)
r = torch.randn(H * Wout * cout, device="cuda").to(torch.bfloat16)
w = torch.from_numpy(L.pack_conv_weight(np.random.randn(cout, cin, k, k).astype(np.float32) * 0.05, dt)).cuda()
sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    L.call("rd_conv2d_bn_act", x.data_ptr(), cs, 0, w.data_ptr(), sc.data_ptr(), sh.data_ptr(), r.data_ptr(), cout, 0,
           y.data_ptr(), cout, 0, 1, H, W, cin, cout, k, k, s, fl, dt, st)
torch.cuda.synchronize()
print("done")
