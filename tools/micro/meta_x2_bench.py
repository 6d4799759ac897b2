"""Dev harness: meta16_kernel (variant 0) against meta16x2_kernel (variant 1) at the production shape -- outputs compared bit for
bit, both timed.  Builds tools/micro/meta_x2.hip into gpurun_out-independent tools/micro/libmeta_x2.so on the fly (hipcc, seconds).
    python tools/micro/meta_x2_bench.py [B] [reps]"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from rangedet_amd import lib as rdlib, synth  # noqa: E402
from rangedet_amd.runtime import TorchAllocator, bn_affine  # noqa: E402

so = os.path.join(HERE, "libmeta_x2.so")
src = os.path.join(HERE, "meta_x2.hip")
if os.environ.get("MX2_FLAGS") is not None or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "rangedet_amd", "csrc", "k_meta.h"))):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result"] +
                          os.environ.get("MX2_FLAGS", "").split() + [src, "-o", so])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
H, W = 64, 2656
L, A = rdlib.get_lib(), TorchAllocator()
M = ctypes.CDLL(so)
M.mx_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
P = synth.make_weights(seed=18)
name, pre = 'res1_unit2', 'res1_unit2_%d' % W
s1, t1 = bn_affine(P, name + "point_wise_mlp_bn1", 1e-5 + 1e-10)
s2, t2 = bn_affine(P, name + "aggregation_bn1", 1e-5 + 1e-10)
pk = A.upload(L.pack_meta(P[pre + "_mlp0_weight"].reshape(32, 3), P[pre + "_mlp0_bias"], P[pre + "_mlp1_weight"].reshape(64, 32),
                          P[pre + "_mlp1_bias"], s1, t1, P[name + "aggregation_conv1_weight"].reshape(64, 576), s2, t2, rdlib.RD_BF16))
x = torch.relu(torch.randn(B, H, W, 64, device="cuda")).to(torch.bfloat16)
c = torch.randn(B, 3, H, W, device="cuda") * 20
ys = [torch.zeros(B, H, W, 64, device="cuda", dtype=torch.bfloat16) for _ in range(2)]
st = torch.cuda.current_stream().cuda_stream


def run(v):
    rc = M.mx_launch(v, x.data_ptr(), 64, 0, c.data_ptr(), A.ptr(pk), ys[v].data_ptr(), 64, 0, B, H, W, st)
    assert rc == 0, rc


for v in (0, 1):
    run(v)
torch.cuda.synchronize()
same = torch.equal(ys[0].view(torch.int16), ys[1].view(torch.int16))
print("outputs bit-identical: %s  (mean |y| %.4f, differing values %d)" % (same, float(ys[0].float().abs().mean()),
                                                                           int((ys[0].view(torch.int16) != ys[1].view(torch.int16)).sum())))
for rnd in range(2):
    for v in (0, 1):
        for _ in range(3):
            run(v)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run(v)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        by = B * H * W * (128 * 2 + 12)
        print("variant %d (%s): %.1f us per launch, %.0f GB/s (%.1f %% of 8 TB/s), %.0f TFLOP/s" % (
            v, "8 waves x 1 fragment" if v == 0 else "4 waves x 2 fragments", us, by / us / 1e3, by / us / 1e3 / 80, B * 19.29e9 / us / 1e6), flush=True)
