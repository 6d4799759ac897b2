"""Dev harness (round 6): meta16_kernel with twelve waves per workgroup / W1 from global memory (tools/micro/meta_w12.hip; variant = 1000 x waves + form) against the shipping 8-wave form at the
production shape -- outputs compared bit for bit, all timed alternately.  Builds tools/micro/meta_w12.hip into
tools/micro/libmeta_w12.so on the fly (hipcc, seconds per form).  MV_FLAGS: extra hipcc flags, e.g. '-DMV_LIST="X(0) X(8)"', '-DMV_DT=RD_F16' (the fp16 instantiations).
    python tools/micro/meta_v_bench.py [B] [reps]"""
import ctypes
import os
import shlex
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from rangedet_amd import lib as rdlib, synth  # noqa: E402
from rangedet_amd.runtime import TorchAllocator, bn_affine  # noqa: E402

so = os.path.join(HERE, "libmeta_w12.so")
src = os.path.join(HERE, "meta_w12.hip")
newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "rangedet_amd", "csrc", "k_meta.h")))
if os.environ.get("MV_FLAGS") is not None or not os.path.exists(so) or os.path.getmtime(so) < newest:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", "-Xarch_device", "-fno-slp-vectorize"] +
                          shlex.split(os.environ.get("MV_FLAGS", "")) + [src, "-o", so])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
H, W = 64, 2656
L, A = rdlib.get_lib(), TorchAllocator()
M = ctypes.CDLL(so)
M.mw_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
DT = rdlib.RD_BF16
tdt = torch.float16 if DT == rdlib.RD_F16 else torch.bfloat16
vbuf = (ctypes.c_int * 64)()
variants = list(vbuf[: M.mw_variants(vbuf, 64)])
P = synth.make_weights(seed=18)
name, pre = 'res1_unit2', 'res1_unit2_%d' % W
s1, t1 = bn_affine(P, name + "point_wise_mlp_bn1", 1e-5 + 1e-10)
s2, t2 = bn_affine(P, name + "aggregation_bn1", 1e-5 + 1e-10)
pk = A.upload(L.pack_meta(P[pre + "_mlp0_weight"].reshape(32, 3), P[pre + "_mlp0_bias"], P[pre + "_mlp1_weight"].reshape(64, 32),
                          P[pre + "_mlp1_bias"], s1, t1, P[name + "aggregation_conv1_weight"].reshape(64, 576), s2, t2, DT))
x = torch.relu(torch.randn(B, H, W, 64, device="cuda")).to(tdt)
c = torch.randn(B, 3, H, W, device="cuda") * 20
ys = {v: torch.zeros(B, H, W, 64, device="cuda", dtype=tdt) for v in variants}
st = torch.cuda.current_stream().cuda_stream


def run(v):
    rc = M.mw_launch(v, x.data_ptr(), 64, 0, c.data_ptr(), A.ptr(pk), ys[v].data_ptr(), 64, 0, B, H, W, st)
    assert rc == 0, (v, rc)


for v in variants:
    run(v)
torch.cuda.synchronize()
ref = ys[variants[0]].view(torch.int16)
for v in variants[1:]:
    d = int((ref != ys[v].view(torch.int16)).sum())
    print("V=%d against V=%d: %s (%d differing values of %d, mean |y| %.4f)" % (
        v, variants[0], "bit-identical" if d == 0 else "DIFFERENT", d, ref.numel(), float(ys[v].float().abs().mean())), flush=True)
best = {v: 1e9 for v in variants}
for rnd in range(3):
    for v in variants:
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
        best[v] = min(best[v], us)
        print("round %d  V=%-3d %.1f us per launch" % (rnd, v, us), flush=True)
by = B * H * W * (128 * 2 + 12)
for v in variants:
    us = best[v]
    print("V=%-3d best %.1f us: %.0f GB/s (%.1f %% of 8 TB/s), %.0f TFLOP/s" % (v, us, by / us / 1e3, by / us / 1e3 / 80, B * 19.29e9 / us / 1e6))
