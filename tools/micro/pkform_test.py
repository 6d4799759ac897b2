"""Which packed-fp32 forms give wrong results next to which co-resident kernel (tools/micro/victim.hip pkform_kernel)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rangedet_amd import lib as R  # noqa: E402

L = R.get_lib()
V = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvictim.so"))
V.pkform_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B, H, W = 8, 64, 2656
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def conv_load(c):
    x = torch.randn(B * H * W * c, device="cuda").to(torch.bfloat16)
    y = torch.empty(B * H * W * c, device="cuda", dtype=torch.bfloat16)
    w = torch.from_numpy(L.pack_conv3x3_ex(np.random.randn(c, c, 3, 3).astype(np.float32) * 0.05, 1, c, fold_scale=np.ones(c, np.float32), dtype=R.RD_BF16)).cuda()
    sh = torch.zeros(c, device="cuda")

    def run():
        for _ in range(12):
            L.call("rd_conv3x3_bn_act_ex", x.data_ptr(), c, 0, w.data_ptr(), None, sh.data_ptr(), None, 0, 0, None, 0, 0, 0, None,
                   y.data_ptr(), c, 0, B, H, W, c, c, 1, R.RD_RELU_POST | R.RD_SCALE_FOLDED, R.RD_BF16, s1.cuda_stream)
    return run, (x, y, w, sh)


MA = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
MB = torch.empty_like(MA)
MF = torch.randn(4096, 4096, device="cuda")
MG = torch.empty_like(MF)


def mm_load():
    with torch.cuda.stream(s1):
        for _ in range(3):
            torch.mm(MA, MA, out=MB)


def mm32_load():
    with torch.cuda.stream(s1):
        for _ in range(6):
            torch.mm(MF, MF, out=MG)


def ew_load():
    with torch.cuda.stream(s1):
        for _ in range(20):
            torch.add(MA, MA, out=MB)


c64 = conv_load(64)
loads = {"none": lambda: None, "conv 64->64": c64[0], "torch.mm bf16 8192^3": mm_load, "torch.mm fp32 4096^3": mm32_load, "torch.add (HBM-bound)": ew_load}
names = ["pk_mul op_sel:[0,1] op_sel_hi:[1,0]", "pk_mul neg only", "pk_add op_sel:[0,1] op_sel_hi:[1,0]", "pk_fma op_sel:[0,1,0] op_sel_hi:[1,0,1]",
         "pk_mov_b32 op_sel:[1,0]", "pk_mul op_sel:[1,0] op_sel_hi:[0,1]", "pk_mul op_sel:[1,1] op_sel_hi:[1,1]", "pk_mul op_sel:[0,0] op_sel_hi:[0,0]",
         "pk_mul op_sel:[0,1] op_sel_hi:[1,1]", "pk_mul (default)"]
cnt = torch.zeros(80, device="cuda", dtype=torch.int32)
for lname, ld in loads.items():
    cnt.zero_()
    torch.cuda.synchronize()
    for r in range(reps):
        ld()
        for pat in range(10):
            V.pkform_run(pat, cnt.data_ptr(), 1024, 1000, s2.cuda_stream)
        torch.cuda.synchronize()
    c = cnt.cpu().numpy().reshape(10, 2, 4)
    print("load %s: mismatches by lane quarter (lo half | hi half)" % lname)
    for p in range(10):
        if c[p].any() or lname == "conv 64->64":
            print("    %-42s %s | %s" % (names[p], c[p, 0].tolist(), c[p, 1].tolist()), flush=True)
