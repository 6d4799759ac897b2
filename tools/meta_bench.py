"""Meta-Kernel alone at the production shape (B frames of 64 x 2656, bf16): average launch time over `reps` launches, GB/s of
compulsory traffic (262 B/px) and TFLOP/s.  RD_META_VARIANT selects the kernel form (rd_api.hip).
    python tools/meta_bench.py [B] [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rangedet_amd import lib as rdlib, synth  # noqa: E402
from rangedet_amd.runtime import TorchAllocator, bn_affine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
H, W = 64, 2656
L, A = rdlib.get_lib(), TorchAllocator()
P = synth.make_weights(seed=18)
name, pre = 'res1_unit2', 'res1_unit2_%d' % W
s1, t1 = bn_affine(P, name + "point_wise_mlp_bn1", 1e-5 + 1e-10)
s2, t2 = bn_affine(P, name + "aggregation_bn1", 1e-5 + 1e-10)
pk = A.upload(L.pack_meta(P[pre + "_mlp0_weight"].reshape(32, 3), P[pre + "_mlp0_bias"], P[pre + "_mlp1_weight"].reshape(64, 32),
                          P[pre + "_mlp1_bias"], s1, t1, P[name + "aggregation_conv1_weight"].reshape(64, 576), s2, t2, rdlib.RD_BF16))
x = (torch.randn(B, H, W, 64, device="cuda") * 1.0).to(torch.bfloat16)
c = torch.randn(B, 3, H, W, device="cuda")
y = torch.empty(B, H, W, 64, device="cuda", dtype=torch.bfloat16)


def run():
    L.call("rd_meta_kernel_fwd", x.data_ptr(), 64, 0, c.data_ptr(), A.ptr(pk), y.data_ptr(), 64, 0, B, H, W, rdlib.RD_BF16, A.stream)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
by = B * H * W * (128 * 2 + 12)
print("meta variant %-5s B=%d: %.1f us per launch, %.0f GB/s (%.1f %% of 8 TB/s), %.0f TFLOP/s, checksum %.4f" % (
    os.environ.get("RD_META_VARIANT", "dflt"), B, us, by / us / 1e3, by / us / 1e3 / 80, B * 19.29e9 / us / 1e6,
    float(y.float().abs().mean())))
