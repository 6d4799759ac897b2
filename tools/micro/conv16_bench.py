"""GPU A/B of the two MFMA shapes of the persistent 3x3 conv (tools/micro/conv16_dev.hip): cout 128 on the 8 x 32 tiles as
v_mfma_f32_32x32x16 (shipped until round 6) and as v_mfma_f32_16x16x32 (M16).  Same inputs (post-ReLU random activations, N(0, 1/sqrt(9 cin))
weights with folded scales), outputs compared with each other and (first case) with torch fp32, times alternated.
    python tools/micro/conv16_bench.py [reps]"""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(HERE, os.environ.get("CONV16_LIB", "libconv16_dev.so")))   # (CONV16_LIB: another build of the harness, for A/B runs)
L.rdm_conv3_packed_bytes.restype = ctypes.c_size_t
L.rdm_last_error.restype = ctypes.c_char_p
vp, ci = ctypes.c_void_p, ctypes.c_int
L.rdm_pack_conv3.argtypes = [vp, vp, ci, ci, ci, ci, vp]
L.rdm_conv3.argtypes = [vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
st = torch.cuda.current_stream().cuda_stream
H, B, cout = 64, 8, 128
RELU, ADD = 4, 2


def pack(w, fs, m16):
    out = np.zeros(L.rdm_conv3_packed_bytes(w.shape[1], w.shape[0]), np.uint8)
    assert L.rdm_pack_conv3(w.ctypes.data, fs.ctypes.data, w.shape[0], w.shape[1], m16, 1, out.ctypes.data) == 0
    return torch.from_numpy(out).cuda()


first = True
for W, cin, add in ((2656, 128, 0), (2656, 128, 1), (1328, 128, 0), (1328, 64, 0), (664, 128, 0), (664, 128, 1), (332, 128, 0), (332, 128, 1), (166, 128, 0), (166, 128, 1)):
    g = torch.Generator(device="cuda").manual_seed(W + cin + add)
    x = torch.relu(torch.randn(B, H, W, cin, device="cuda", generator=g)).to(torch.bfloat16)
    r = torch.randn(B, H, W, cout, device="cuda", generator=g).to(torch.bfloat16)
    rng = np.random.default_rng(1)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    fs = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).cuda()
    wp = [pack(w, fs, 0), pack(w, fs, 1)]
    y = [torch.empty(B, H, W, cout, device="cuda", dtype=torch.bfloat16) for _ in range(2)]

    def run(m):
        rc = L.rdm_conv3(x.data_ptr(), cin, wp[m].data_ptr(), sh.data_ptr(), r.data_ptr() if add else None, y[m].data_ptr(), B, H, W, cin, cout,
                         RELU | (ADD if add else 0), m, 1, st)
        assert rc == 0, L.rdm_last_error()
    run(0), run(1)
    torch.cuda.synchronize()
    d = (y[0].float() - y[1].float()).abs()
    msg = "max |32x32 - 16x16| %.4f (%.4f %% of the values differ)" % (d.max().item(), 100.0 * (d > 0).float().mean().item())
    if first:   # against torch fp32 on one frame
        wt = torch.from_numpy(w * fs[:, None, None, None]).cuda().to(torch.bfloat16).float()
        ref = torch.relu(torch.nn.functional.conv2d(x[:1].float().permute(0, 3, 1, 2), wt, padding=1) + sh[None, :, None, None]).permute(0, 2, 3, 1)
        for m in (0, 1):
            e = (y[m][:1].float() - ref).abs()
            msg += "; form %d vs torch fp32 (bf16 weights): max %.4f, > 1 bf16 ulp on %.5f %%" % (m, e.max().item(), 100.0 * (e > ref.abs() * 2.0 ** -7 + 1e-3).float().mean().item())
        first = False
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
    fl = 2.0 * B * H * W * cin * cout * 9
    print("W %4d cin %3d %s  32x32x16 %s us (%.0f TFLOP/s)   16x16x32 %s us (%.0f TFLOP/s)   %s" % (
        W, cin, "+res" if add else "    ", "/".join("%.1f" % v for v in t[0]), fl / min(t[0]) / 1e6, "/".join("%.1f" % v for v in t[1]), fl / min(t[1]) / 1e6, msg), flush=True)

# ---- the M16 form with the fused 1x1 output conv (16-row MFMAs on the registers the epilogue has just rounded) ------------------------------
L.rdm_pack_head16.argtypes = [vp, ci, ci, ci, vp]
L.rdm_conv3_head.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
for W, nout in ((2656, 8), (2656, 1), (1328, 8), (664, 8)):
    cin = 128
    g = torch.Generator(device="cuda").manual_seed(W + nout)
    x = torch.relu(torch.randn(B, H, W, cin, device="cuda", generator=g)).to(torch.bfloat16)
    rng = np.random.default_rng(2)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    fs = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).cuda()
    hw = (rng.standard_normal((nout, 128)) / np.sqrt(128)).astype(np.float32)
    hb = torch.from_numpy(rng.standard_normal(8).astype(np.float32)).cuda()
    hp = np.zeros(8192, np.uint8)
    assert L.rdm_pack_head16(hw.ctypes.data, nout, 128, 1, hp.ctypes.data) == 0
    hp = torch.from_numpy(hp).cuda()
    wp16 = pack(w, fs, 1)
    out = torch.zeros(B, H * W, nout, device="cuda")
    y16 = torch.empty(B, H, W, cout, device="cuda", dtype=torch.bfloat16)

    def run_h():
        rc = L.rdm_conv3_head(x.data_ptr(), cin, wp16.data_ptr(), sh.data_ptr(), hp.data_ptr(), hb.data_ptr(), out.data_ptr(), nout, B, H, W, cin, 1, st)
        assert rc == 0, L.rdm_last_error()

    def run_p():
        assert L.rdm_conv3(x.data_ptr(), cin, wp16.data_ptr(), sh.data_ptr(), None, y16.data_ptr(), B, H, W, cin, cout, RELU, 1, 1, st) == 0
    run_h(), run_p()
    torch.cuda.synchronize()
    want = y16[:1].float().reshape(1, H * W, cout) @ torch.from_numpy(hw).cuda().T + hb[:nout]
    err = (out[:1] - want).abs().max().item()
    t = [[], []]
    for rep in range(3):
        for m, fn in enumerate((run_p, run_h)):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t[m].append(e0.elapsed_time(e1) * 1e3 / reps)
    print("W %4d nout %d: plain 16x16x32 conv %s us   + fused output conv %s us   max |out - head(conv)| %.2e (spread %.2f)" % (
        W, nout, "/".join("%.1f" % v for v in t[0]), "/".join("%.1f" % v for v in t[1]), err, want.std().item()), flush=True)
