"""Negative control of the emulator's late-DMA mode (tests/emu/hip/hip_runtime.h dma_late; TEST INFRASTRUCTURE, CPU only).
The persistent conv's "no counted wait" ablation (RD_CONV3_DBG=32 of a -DRD_CONV3_DEV build: every s_waitcnt vmcnt(N) becomes vmcnt(63)) must
give the right answer when LDS-DMA transfers land at issue -- the only model the emulator had until round 5 -- and a WRONG one when they land
as late as the waits allow; the production build must be right in both.  Result of round 5: profiles/r05h_emu_late_dma.txt.
    /opt/rocm/lib/llvm/bin/clang++ -x c++ -std=c++17 -O2 -fPIC -shared -DRD_CONV3_DEV -DRD_BUILD_NUM_CUS=4 -DRD_BUILD_F16_PRODUCTION_FORMS_ONLY \
        -Itests/emu -Iinclude rangedet_amd/csrc/rd_api.hip -o /tmp/emu_dev.so          (10 min)
    python tools/emu_late_dma_control.py /tmp/emu_dev.so"""
import os, sys
import numpy as np
os.environ["RD_CONV3_DBG"] = "32"
DEV_SO = sys.argv[1] if len(sys.argv) > 1 else "/tmp/emu_dev.so"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rangedet_amd import lib as R
from emu_util import NumpyAllocator, f32_to_bf16_bits
A = NumpyAllocator()
B, H, W, c = 1, 16, 100, 128
rng = np.random.default_rng(0)
x = f32_to_bf16_bits(rng.standard_normal((B, H, W, c)).astype(np.float32))
wt = rng.standard_normal((c, c, 3, 3)).astype(np.float32) * 0.05
outs = {}
for tag, path in (("dev build, counted waits removed", DEV_SO), ("production build", os.path.join(ROOT, "tests", "emu", "librangedet_emu.so"))):
    L = R.Lib(path)
    wp = L.pack_conv3x3_ex(wt, 1, c, fold_scale=np.ones(c, np.float32), dtype=R.RD_BF16)
    dx, dw, sh = A.upload(x), A.upload(wp), A.upload(np.zeros(c, np.float32))
    for late in (0, 1):
        L.cdll.hipemu_set_dma_late(late)
        y = A.alloc(B * H * W * c * 2, zero=True)
        L.call("rd_conv3x3_bn_act_ex", A.ptr(dx), c, 0, A.ptr(dw), None, A.ptr(sh), None, 0, 0, None, 0, 0, 0, None, A.ptr(y), c, 0, B, H, W, c, c, 1,
               R.RD_RELU_POST | R.RD_SCALE_FOLDED, R.RD_BF16, A.stream)
        outs[(tag, late)] = np.array(y[: B * H * W * c * 2]).view(np.uint16).copy()
    L.cdll.hipemu_set_dma_late(0)
ref = outs[("production build", 0)]
for k, v in outs.items():
    print("%-34s transfers land %-8s: %6d of %d output values differ from the production build / at issue" % (k[0], "LATE" if k[1] else "at issue", int((v != ref).sum()), v.size))
