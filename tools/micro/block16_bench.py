"""GPU A/B of the two MFMA shapes of the fused 64-channel BasicBlock kernel (tools/micro/block16_dev.hip): outputs compared bit for bit, times alternated.
    python tools/micro/block16_bench.py [reps]"""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(HERE, "libblock16_dev.so"))
L.rdm_block_packed_bytes.restype = ctypes.c_size_t
L.rdm_last_error.restype = ctypes.c_char_p
vp, ci = ctypes.c_void_p, ctypes.c_int
L.rdm_pack_block.argtypes = [vp, vp, vp, vp, ci, vp]
L.rdm_pack_sc.argtypes = [vp, vp, ci, vp]
L.rdm_block.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
st = torch.cuda.current_stream().cuda_stream
for B, H, W, sc in ((2, 11, 70, 0), (2, 11, 70, 1), (8, 64, 2656, 0), (8, 64, 2656, 1), (8, 64, 1328, 0), (8, 64, 1328, 1)):
    g = torch.Generator(device="cuda").manual_seed(W + sc)
    x = torch.relu(torch.randn(B, H, W, 64, device="cuda", generator=g)).to(torch.bfloat16)
    rng = np.random.default_rng(3)
    w1, w2 = ((rng.standard_normal((64, 64, 3, 3)) / 24).astype(np.float32) for _ in range(2))
    wsc = (rng.standard_normal((64, 64)) / 8).astype(np.float32)
    s1, s2, ss = (rng.uniform(0.5, 1.5, 64).astype(np.float32) for _ in range(3))
    t1, t2 = (torch.from_numpy((rng.standard_normal(64) * 0.3).astype(np.float32)).cuda() for _ in range(2))
    pk, psc = [], []
    for m in (0, 1):
        o = np.zeros(L.rdm_block_packed_bytes(), np.uint8)
        L.rdm_pack_block(w1.ctypes.data, s1.ctypes.data, w2.ctypes.data, s2.ctypes.data, m, o.ctypes.data)
        pk.append(torch.from_numpy(o).cuda())
        o = np.zeros(8192, np.uint8)
        L.rdm_pack_sc(wsc.ctypes.data, ss.ctypes.data, m, o.ctypes.data)
        psc.append(torch.from_numpy(o).cuda())
    y = [torch.empty(B, H, W, 64, device="cuda", dtype=torch.bfloat16) for _ in range(2)]

    def run(m):
        rc = L.rdm_block(x.data_ptr(), pk[m].data_ptr(), t1.data_ptr(), t2.data_ptr(), psc[m].data_ptr() if sc else None, y[m].data_ptr(), B, H, W, m, st)
        assert rc == 0, L.rdm_last_error()
    run(0), run(1)
    torch.cuda.synchronize()
    ndiff = int((y[0].view(torch.int16) != y[1].view(torch.int16)).sum().item())
    t = [[], []]
    for rep in range(3):
        for m in (0, 1):
            for _ in range(3):
                run(m)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run(m)
            e1.record()
            torch.cuda.synchronize()
            t[m].append(e0.elapsed_time(e1) * 1e3 / reps)
    print("B %d H %2d W %4d %s  32x32x16 %s us   16x16x32 %s us   %d of %d output values differ (|y| mean %.3f)" % (
        B, H, W, "+sc" if sc else "   ", "/".join("%.1f" % v for v in t[0]), "/".join("%.1f" % v for v in t[1]), ndiff, y[0].numel(), y[0].float().abs().mean().item()), flush=True)
