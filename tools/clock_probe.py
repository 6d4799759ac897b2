"""Shader clock / board power (hwmon files, tools/power_sample.py) while ONE production launch form loops for ~2 s each: which launches sit at the
power cap (low clock at full power) and which wait on memory (clock near the 2.4-GHz maximum).  Post-ReLU random inputs, 8 frames, 64 rows.
    python tools/clock_probe.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from rangedet_amd import lib as R  # noqa: E402
from power_sample import PowerSampler  # noqa: E402

L = R.get_lib()
st = torch.cuda.current_stream().cuda_stream
B, H, dt = 8, 64, R.RD_BF16
rng = np.random.default_rng(0)


def loop(name, fn, flops):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    with PowerSampler(period=0.01) as ps:
        t0 = time.time()
        n = 0
        while time.time() - t0 < 2.0:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            n += 50
        dtm = time.time() - t0
    print("%-44s %7.1f us  %6.0f TFLOP/s   %s" % (name, dtm / n * 1e6, flops * n / dtm / 1e12, ps.summary()), flush=True)


def act(C, W):
    return torch.relu(torch.randn(B, H, W, C, device="cuda")).to(torch.bfloat16)


for W in (2656, 1328):
    # tower conv 128 -> 128 in both MFMA shapes
    w = (rng.standard_normal((128, 128, 3, 3)) / np.sqrt(9 * 128)).astype(np.float32)
    fs = rng.uniform(0.5, 1.5, 128).astype(np.float32)
    sh = torch.zeros(128, device="cuda")
    x, y = act(128, W), torch.empty(B, H, W, 128, device="cuda", dtype=torch.bfloat16)
    for m16 in (0, 1):
        wp = torch.from_numpy(L.pack_conv3x3_m16(w, fs, dt) if m16 else L.pack_conv3x3_ex(w, 1, 128, fold_scale=fs, dtype=dt)).cuda()
        fl = R.RD_RELU_POST | R.RD_SCALE_FOLDED | (R.RD_MFMA16 if m16 else 0)
        loop("conv 128->128 W%d %s" % (W, "16x16x32" if m16 else "32x32x16"),
             lambda: L.call("rd_conv3x3_bn_act_ex", x.data_ptr(), 128, 0, wp.data_ptr(), None, sh.data_ptr(), None, 0, 0, None, 0, 0, 0, None,
                            y.data_ptr(), 128, 0, B, H, W, 128, 128, 1, fl, dt, st), 2.0 * B * H * W * 128 * 128 * 9)
    # fused 64-channel BasicBlock and the plain 64 -> 64 conv
    w1 = (rng.standard_normal((64, 64, 3, 3)) / np.sqrt(9 * 64)).astype(np.float32)
    s64 = rng.uniform(0.5, 1.5, 64).astype(np.float32)
    wb = torch.from_numpy(L.pack_block64(w1, s64, w1, s64, dt)).cuda()
    sh64 = torch.zeros(64, device="cuda")
    x6, y6 = act(64, W), torch.empty(B, H, W, 64, device="cuda", dtype=torch.bfloat16)
    loop("block 64->64->64 W%d" % W,
         lambda: L.call("rd_block64_bn_act", x6.data_ptr(), 64, 0, 64, wb.data_ptr(), sh64.data_ptr(), sh64.data_ptr(), None, y6.data_ptr(), 64, 0,
                        B, H, W, dt, st), 2.0 * 2 * B * H * W * 64 * 64 * 9)
    wc = torch.from_numpy(L.pack_conv3x3_ex(w1, 1, 64, fold_scale=s64, dtype=dt)).cuda()
    loop("conv 64->64 W%d" % W,
         lambda: L.call("rd_conv3x3_bn_act_ex", x6.data_ptr(), 64, 0, wc.data_ptr(), None, sh64.data_ptr(), None, 0, 0, None, 0, 0, 0, None,
                        y6.data_ptr(), 64, 0, B, H, W, 64, 64, 1, R.RD_RELU_POST | R.RD_SCALE_FOLDED, dt, st), 2.0 * B * H * W * 64 * 64 * 9)

# the fused Meta-Kernel unit at the production shape
from rangedet_amd import synth  # noqa: E402
from rangedet_amd.runtime import TorchAllocator, bn_affine  # noqa: E402
A = TorchAllocator()
P = synth.make_weights(seed=18)
name, pre = 'res1_unit2', 'res1_unit2_%d' % 2656
s1, t1 = bn_affine(P, name + "point_wise_mlp_bn1", 1e-5 + 1e-10)
s2, t2 = bn_affine(P, name + "aggregation_bn1", 1e-5 + 1e-10)
pk = A.upload(L.pack_meta(P[pre + "_mlp0_weight"].reshape(32, 3), P[pre + "_mlp0_bias"], P[pre + "_mlp1_weight"].reshape(64, 32),
                          P[pre + "_mlp1_bias"], s1, t1, P[name + "aggregation_conv1_weight"].reshape(64, 576), s2, t2, dt))
xm, cm = act(64, 2656), torch.randn(B, 3, H, 2656, device="cuda")
ym = torch.empty(B, H, 2656, 64, device="cuda", dtype=torch.bfloat16)
loop("meta unit W2656", lambda: L.call("rd_meta_kernel_fwd", xm.data_ptr(), 64, 0, cm.data_ptr(), A.ptr(pk), ym.data_ptr(), 64, 0, B, H, 2656, dt, A.stream), B * 19.29e9)
