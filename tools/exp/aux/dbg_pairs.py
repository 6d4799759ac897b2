"""debug: where rd_deconv2d_bn_act_pairs differs from rd_deconv2d_bn_act_all"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))), "tests"))
import torch
from rangedet_amd import lib as R
from rangedet_amd.runtime import TorchAllocator
L, A = R.get_lib(), TorchAllocator()
dt = R.RD_BF16
B, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
USE_SH, USE_RES = int(sys.argv[4]), int(sys.argv[5])
cin, cout, k, s, pw = 128, 64, (3, 8), 4, 2
rng = np.random.default_rng(3)
x = rng.standard_normal((B, H, W, cin)).astype(np.float32)
w = (rng.standard_normal((cin, cout, k[0], k[1])) / 20).astype(np.float32)
sc = np.ones(cout, np.float32); sh = rng.standard_normal(cout).astype(np.float32) if USE_SH else np.zeros(cout, np.float32)
xin = torch.from_numpy(x).to(torch.bfloat16).cuda()
imgs = [L.pack_deconv_weight(w, s, pw, ph, dt, fold_scale=sc) for ph in range(s)]
pairs = [L.pack_deconv_phase_pair(imgs[p], imgs[p + 1], cin, dt) for p in (0, 2)]
Wout = s * W
y1 = torch.zeros(B, H, Wout, cout, dtype=torch.bfloat16, device="cuda"); y2 = torch.zeros_like(y1)
fl = (R.RD_RELU_PRE | R.RD_ADD | R.RD_SCALE_FOLDED) if USE_RES else (R.RD_RELU_POST | R.RD_SCALE_FOLDED)
res = torch.randn(B, H, Wout, cout, device='cuda').to(torch.bfloat16)
rp = res.data_ptr() if USE_RES else None
wa, wp = A.upload(np.concatenate(imgs)), A.upload(np.concatenate(pairs))
dsh, dsh2 = A.upload(sh), A.upload(np.concatenate([sh, sh]))
L.call("rd_deconv2d_bn_act_all", xin.data_ptr(), cin, 0, A.ptr(wa), len(imgs[0]), A.ptr(dsh), rp, cout, 0, y1.data_ptr(), cout, 0, B, H, W, cin, cout, k[0], k[1], s, pw, fl, dt, A.stream)
L.call("rd_deconv2d_bn_act_pairs", xin.data_ptr(), cin, 0, A.ptr(wp), len(pairs[0]), A.ptr(dsh2), rp, cout, 0, y2.data_ptr(), cout, 0, B, H, W, cin, cout, k[0], k[1], s, pw, fl, dt, A.stream)
torch.cuda.synchronize()
d = (y1.view(torch.int16) != y2.view(torch.int16)).cpu().numpy()
print("differ", d.sum(), "of", d.size)
dd = d.reshape(B, H, W, s, cout)
print("by phase", dd.sum(axis=(0, 1, 2, 4)))
print("by channel block of 8", dd.reshape(B, H, W, s, 8, 8).sum(axis=(0, 1, 2, 3, 5)))
print("by row", dd.sum(axis=(0, 2, 3, 4)))
print("by col", dd.sum(axis=(0, 1, 3, 4)))
a = y1.float().cpu().numpy().reshape(B, H, W, s, cout); b = y2.float().cpu().numpy().reshape(B, H, W, s, cout)
i = np.argwhere(dd)[:5]
for t in i: print(t, a[tuple(t)], b[tuple(t)])
import torch.nn.functional as F
xr = xin.float().cpu().permute(0, 3, 1, 2)
wr = torch.from_numpy(w).to(torch.bfloat16).float()
ref = F.conv_transpose2d(xr, wr, stride=(1, s), padding=(1, pw)) + torch.from_numpy(sh)[None, :, None, None]
ref = torch.relu(ref)
if USE_RES: ref = ref + res.float().cpu().permute(0, 3, 1, 2)
ref = ref.permute(0, 2, 3, 1).numpy().reshape(B, H, W, s, cout)
for nm, v in (("all", a), ("pairs", b)):
    e = np.abs(v - ref)
    print(nm, "max err", e.max(), "rows with err > 0.1:", np.unique(np.argwhere(e > 0.1)[:, 1]), "count", int((e > 0.1).sum()))
