"""Co-residency study: plain victim kernels (tools/micro/victim.hip) next to a 64->64 / 128->128 3x3 conv of the library.
    python tools/micro/victim_test.py [reps]"""
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
V.victim_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, H, W = 8, 64, 2656
DT = R.RD_BF16
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
NBLK, ITERS = int(os.environ.get("NBLK", "2048")), int(os.environ.get("ITERS", "40000"))
out = torch.empty(NBLK * 256, device="cuda")


def conv_load(c):
    x = torch.randn(B * H * W * c, device="cuda").to(torch.bfloat16)
    y = torch.empty(B * H * W * c, device="cuda", dtype=torch.bfloat16)
    w = torch.from_numpy(L.pack_conv3x3_ex(np.random.randn(c, c, 3, 3).astype(np.float32) * 0.05, 1, c, fold_scale=np.ones(c, np.float32), dtype=DT)).cuda()
    sh = torch.zeros(c, device="cuda")
    fl = R.RD_RELU_POST | R.RD_SCALE_FOLDED

    def run(st):
        L.call("rd_conv3x3_bn_act_ex", x.data_ptr(), c, 0, w.data_ptr(), None, sh.data_ptr(), None, 0, 0, None, 0, 0, 0, None,
               y.data_ptr(), c, 0, B, H, W, c, c, 1, fl, DT, st)
    return run, (x, y, w, sh)


loads = {"none": None, "conv64": conv_load(64), "conv128": conv_load(128)}
for mode in range(4):
    V.victim_run(mode, out.data_ptr(), NBLK, ITERS, 1.0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    base = out.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); V.victim_run(mode, out.data_ptr(), NBLK, ITERS, 1.0, torch.cuda.current_stream().cuda_stream); e1.record()
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int32), base.view(torch.int32)), "idle run not reproducible"
    line = "victim mode %d (%.0f us idle):" % (mode, e0.elapsed_time(e1) * 1e3)
    for name, ld in loads.items():
        nbad = nel = 0
        first = None
        for r in range(reps):
            out.zero_()
            torch.cuda.synchronize()
            if ld:
                for _ in range(12):
                    ld[0](s1.cuda_stream)
            V.victim_run(mode, out.data_ptr(), NBLK, ITERS, 1.0, s2.cuda_stream)
            torch.cuda.synchronize()
            d = out.view(torch.int32) != base.view(torch.int32)
            n = int(d.sum())
            if n:
                nbad += 1
                nel += n
                if first is None:
                    ii = torch.nonzero(d).flatten()[:24].tolist()
                    first = ii
        line += "  %s: %d/%d runs, %d elements%s" % (name, nbad, reps, nel, (" first " + str(first)) if first else "")
    print(line, flush=True)

# ---- write-after-read probes -------------------------------------------------------------------------------------------------
V.war_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
cnt = torch.zeros(20, device="cuda", dtype=torch.int32)
names = ["v_pk_mov_b32", "v_pk_mul_f32", "v_mul_f32 (control)", "v_pk_add_f32", "v_mov_b64"]
for name, ld in loads.items():
    cnt.zero_()
    torch.cuda.synchronize()
    for r in range(reps):
        if ld:
            for _ in range(12):
                ld[0](s1.cuda_stream)
        for pat in range(5):
            V.war_run(pat, cnt.data_ptr(), 1024, 2000, s2.cuda_stream)
        torch.cuda.synchronize()
    c = cnt.cpu().numpy().reshape(5, 4)
    print("WAR probes, load %-8s (mismatches by lane quarter): %s" % (name, "  ".join("%s %s" % (names[p], c[p].tolist()) for p in range(5))), flush=True)

# ---- the packed-fp32 sequence of wnms_prep_kernel ------------------------------------------------------------------------------
V.pk_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
cnt = torch.zeros(48, device="cuda", dtype=torch.int32)
pk_names = ["mov hi; pk_mul; pk_mul op_sel", "mov hi; pk_mul op_sel", "mov hi; s_nop 1; pk_mul op_sel", "mov hi; pk_mul; pk_mul (no op_sel)",
            "mov_b64 pair; pk_mul; pk_mul op_sel", "mov lo; pk_mul; pk_mul op_sel"]
for name, ld in loads.items():
    cnt.zero_()
    torch.cuda.synchronize()
    for r in range(reps):
        if ld:
            for _ in range(12):
                ld[0](s1.cuda_stream)
        for pat in range(6):
            V.pk_run(pat, cnt.data_ptr(), 1024, 2000, s2.cuda_stream)
        torch.cuda.synchronize()
    c = cnt.cpu().numpy().reshape(6, 2, 4)
    print("packed sequence probes, load %s (mismatches by lane quarter, lo half | hi half):" % name)
    for p in range(6):
        print("    %-40s %s | %s" % (pk_names[p], c[p, 0].tolist(), c[p, 1].tolist()), flush=True)
