"""Winograd F(2,3) 3x3 conv experiment (tools/micro/k_wino.h) through its development harness tools/micro/wino_dev.hip.
    python tools/micro/wino_dev.py emu             parity on the CPU emulator (small shapes)
    python tools/micro/wino_dev.py gpu [bench]     parity at full size on the GPU against the production direct kernel, and timing
Checks: (1) against a numpy model of the SAME arithmetic (V and U rounded to the 16-bit type, fp64 accumulation, one output rounding):
at most one 16-bit ulp apart, almost everywhere equal; (2) against the exact conv of the same 16-bit inputs with UNROUNDED weights: the
error model of profiles/EXPERIMENTS.md (round 6).
"""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rangedet_amd import lib as R  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "emu"
c_int, c_void_p = ctypes.c_int, ctypes.c_void_p


def bind(path):
    m = ctypes.CDLL(path)
    m.rdm_wino_packed_bytes.restype = ctypes.c_size_t
    m.rdm_wino_packed_bytes.argtypes = [c_int]
    m.rdm_pack_wino_host.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p]
    m.rdm_wino.argtypes = [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    m.rdm_last_error.restype = ctypes.c_char_p
    return m


def rnd16(a, dt):
    from emu_util import h16_round
    return h16_round(np.asarray(a, np.float32), dt).astype(np.float64)


def bits16(a, dt):
    from emu_util import f32_to_bf16_bits, f32_to_f16_bits
    return f32_to_bf16_bits(a) if dt == R.RD_BF16 else f32_to_f16_bits(a)


def from_bits(b, dt):
    from emu_util import bf16_bits_to_f32, f16_bits_to_f32
    return bf16_bits_to_f32(b) if dt == R.RD_BF16 else f16_bits_to_f32(b)


def wino_model(x, w, scale, shift, res, flags, dt):
    """x (B,H,W,C) float64 values of the 16-bit type; w (128,C,3,3) f32; -> the kernel's arithmetic in fp64 accumulation, before the output rounding."""
    B, H, W, C = x.shape
    ws = (w * scale[:, None, None, None]).astype(np.float32)
    g0, g1, g2 = ws[..., 0], ws[..., 1], ws[..., 2]                 # (co, ci, dh)
    U = [g0, np.float32(0.5) * ((g0 + g2) + g1), np.float32(0.5) * ((g0 + g2) - g1), -g2]
    U = [rnd16(u, dt) for u in U]
    P = (W + 1) // 2
    xp = np.zeros((B, H + 2, 2 * P + 2, C))
    xp[:, 1:H + 1, 1:W + 1] = x
    d = [xp[:, :, j:j + 2 * P:2] for j in range(4)]                   # (B, H+2, P, C)
    V = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[3] - d[1]]
    V = [rnd16(v, dt) for v in V]
    M = []
    for mi in range(4):
        acc = 0
        for dh in range(3):
            acc = acc + np.einsum("bhpc,oc->bhpo", V[mi][:, dh:dh + H], U[mi][:, :, dh])
        M.append(acc)
    y = np.zeros((B, H, 2 * P, 128))
    y[:, :, 0::2] = M[0] + M[1] + M[2]
    y[:, :, 1::2] = M[1] - M[2] - M[3]
    y = y[:, :, :W] + shift
    if flags & R.RD_ADD:
        y = y + res
    if flags & R.RD_RELU_POST:
        y = np.maximum(y, 0)
    return y


def direct_exact(x, w, scale, shift, res, flags):
    B, H, W, C = x.shape
    ws = (w * scale[:, None, None, None]).astype(np.float64)
    xp = np.zeros((B, H + 2, W + 2, C))
    xp[:, 1:H + 1, 1:W + 1] = x
    y = np.zeros((B, H, W, 128))
    for dh in range(3):
        for dw in range(3):
            y += np.einsum("bhwc,oc->bhwo", xp[:, dh:dh + H, dw:dw + W], ws[:, :, dh, dw])
    y = y + shift
    if flags & R.RD_ADD:
        y = y + res
    if flags & R.RD_RELU_POST:
        y = np.maximum(y, 0)
    return y


def ulp16(v, dt):
    e = np.floor(np.log2(np.maximum(np.abs(v), 1e-30)))
    return 2.0 ** (e - (7 if dt == R.RD_BF16 else 10))


def make_case(B, H, W, cin, dt, flags, seed):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((128, cin, 3, 3)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, 128).astype(np.float32)
    shift = (rng.standard_normal(128) * 0.2).astype(np.float32)
    xf = np.maximum(rng.standard_normal((B, H, W, cin)), 0).astype(np.float32)
    rf = rng.standard_normal((B, H, W, 128)).astype(np.float32)
    return w, scale, shift, xf, rf


def run_case(m, alloc, B, H, W, cin, dt, flags, seed, verbose=True):
    w, scale, shift, xf, rf = make_case(B, H, W, cin, dt, flags, seed)
    xb, rb = bits16(xf, dt), bits16(rf, dt)
    pk = np.zeros(m.rdm_wino_packed_bytes(cin), np.uint8)
    m.rdm_pack_wino_host(w.ctypes.data, scale.ctypes.data, cin, dt, pk.ctypes.data)
    y = np.full((B, H, W, 128), 0x7fc0, np.uint16)
    rc = m.rdm_wino(xb.ctypes.data, cin, 0, pk.ctypes.data, shift.ctypes.data, rb.ctypes.data if flags & R.RD_ADD else None, 128, 0,
                    y.ctypes.data, 128, 0, B, H, W, cin, flags | R.RD_SCALE_FOLDED, dt, None)
    assert rc == 0, m.rdm_last_error()
    x64, r64 = from_bits(xb, dt).astype(np.float64), from_bits(rb, dt).astype(np.float64)
    # the bf16 hi + lo pair that carries the shift into the accumulators is exact to 2^-17: use the exact shift in both models
    ym = wino_model(x64, w, scale, shift.astype(np.float64), r64, flags, dt)
    yk = from_bits(y, dt).astype(np.float64)
    ymr = rnd16(ym, dt)
    u = ulp16(np.maximum(np.abs(ym), 0.05), dt)   # (fp32 accumulation order: absolute noise ~1e-6 decides the rounding of values near 0)
    diff = np.abs(yk - ymr)
    nbad = int((diff > 1.001 * u).sum())
    neq = float((yk == ymr).mean())
    yd = direct_exact(x64, w, scale, shift.astype(np.float64), r64, flags)
    rms = np.sqrt((yd ** 2).mean())
    e_w = np.sqrt(((yk - yd) ** 2).mean()) / rms
    if verbose:
        print("B %d H %d W %d cin %d dt %d flags %d: vs model: %d of %d beyond one ulp, %.4f equal; rel rms error vs exact conv %.3e (max %.3e of rms)"
              % (B, H, W, cin, dt, flags, nbad, yk.size, neq, e_w, np.abs(yk - yd).max() / rms), flush=True)
    if nbad or neq < 0.97:
        idx = np.argwhere(diff > 1.001 * u)
        print("  first mismatches (b, h, w, c):", idx[:10].tolist())
        if len(idx):
            print("  rows:", sorted(set(idx[:, 1].tolist()))[:24], "cols:", sorted(set(idx[:, 2].tolist()))[:40], "ch:", sorted(set(idx[:, 3].tolist()))[:20])
        sys.exit(1)


if mode == "emu":
    so = "/tmp/libwino_emu.so"
    srcs = [os.path.join(ROOT, "tools/micro/k_wino.h"), os.path.join(ROOT, "tools/micro/wino_dev.hip"), os.path.join(ROOT, "tests/emu/hip/hip_runtime.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call("/opt/rocm/lib/llvm/bin/clang++ -x c++ -std=c++17 -O2 -fPIC -shared -DRD_BUILD_NUM_CUS=4 -DRD_BUILD_F16_PRODUCTION_FORMS_ONLY "
                              "-Wno-undefined-inline -Itests/emu -Iinclude tools/micro/wino_dev.hip -o %s 2>/dev/null" % so, shell=True, cwd=ROOT)
    m = bind(so)
    late = int(os.environ.get("HIPEMU_DMA_LATE", "0"))
    if late:
        m.hipemu_set_dma_late(1)
    cases = [(1, 4, 64, 32, R.RD_BF16, R.RD_RELU_POST), (2, 8, 130, 64, R.RD_BF16, R.RD_RELU_POST), (1, 11, 33, 128, R.RD_BF16, R.RD_ADD | R.RD_RELU_POST),
             (3, 6, 200, 32, R.RD_F16, R.RD_ADD | R.RD_RELU_POST), (1, 16, 166, 128, R.RD_BF16, 0)]
    for i, (B, H, W, cin, dt, fl) in enumerate(cases):
        run_case(m, None, B, H, W, cin, dt, fl, 100 + i)
    print("emu OK (dma late %d)" % late)
else:
    import torch
    so = os.environ.get("WINO_SO", os.path.join(ROOT, "tools/micro/libwino_dev.so"))
    m, L = bind(so), R.get_lib()
    st = torch.cuda.current_stream().cuda_stream
    dev = "cuda"
    bench = len(sys.argv) > 2
    shapes = ((8, 64, 2656, 128),) if os.environ.get("WINO_DBG") else ((8, 64, 2656, 128), (8, 64, 1328, 128), (8, 64, 664, 128), (8, 64, 332, 128), (8, 64, 166, 128), (3, 64, 2650, 64), (1, 21, 77, 32))
    for dt, tdt in ((R.RD_BF16, torch.bfloat16),) if os.environ.get("WINO_DBG") else ((R.RD_BF16, torch.bfloat16), (R.RD_F16, torch.float16)):
        for (B, H, W, cin) in shapes:
            for flags in (R.RD_RELU_POST, R.RD_ADD | R.RD_RELU_POST):
                if flags & R.RD_ADD and not (W in (664, 77)):
                    continue
                w, scale, shift, _, _ = make_case(1, 1, 1, cin, dt, flags, B + W)
                NB = 3
                g = torch.Generator(device=dev).manual_seed(W + cin)
                xs = [torch.randn(B, H, W, cin, device=dev, generator=g).relu().to(tdt) for _ in range(NB)]
                rs = [torch.randn(B, H, W, 128, device=dev, generator=g).to(tdt) for _ in range(NB)]
                pw = np.zeros(m.rdm_wino_packed_bytes(cin), np.uint8)
                m.rdm_pack_wino_host(w.ctypes.data, scale.ctypes.data, cin, dt, pw.ctypes.data)
                pw = torch.from_numpy(pw).cuda()
                pd = torch.from_numpy(L.pack_conv3x3_ex(w, 1, cin, fold_scale=scale, dtype=dt)).cuda()
                T = torch.from_numpy(shift).cuda()
                yd = [torch.empty(B, H, W, 128, device=dev, dtype=tdt) for _ in range(NB)]
                yw = [torch.full((B, H, W, 128), float("nan"), device=dev, dtype=tdt) for _ in range(NB)]

                def direct(i):
                    L.call("rd_conv3x3_bn_act_ex", xs[i].data_ptr(), cin, 0, pd.data_ptr(), None, T.data_ptr(), rs[i].data_ptr() if flags & R.RD_ADD else None, 128, 0,
                           None, 0, 0, 0, None, yd[i].data_ptr(), 128, 0, B, H, W, cin, 128, 1, flags | R.RD_SCALE_FOLDED, dt, st)

                def wino(i):
                    rc = m.rdm_wino(xs[i].data_ptr(), cin, 0, pw.data_ptr(), T.data_ptr(), rs[i].data_ptr() if flags & R.RD_ADD else None, 128, 0,
                                    yw[i].data_ptr(), 128, 0, B, H, W, cin, flags | R.RD_SCALE_FOLDED, dt, st)
                    assert rc == 0, m.rdm_last_error()
                for i in range(NB):
                    direct(i)
                    wino(i)
                torch.cuda.synchronize()
                # fp32 reference of the same 16-bit inputs with unrounded (scale-folded) weights
                wt = torch.from_numpy(w * scale[:, None, None, None]).cuda()
                stats = []
                for i in range(1):
                    ref = torch.nn.functional.conv2d(xs[i].float().permute(0, 3, 1, 2), wt, padding=1).permute(0, 2, 3, 1) + T
                    if flags & R.RD_ADD:
                        ref = ref + rs[i].float()
                    ref = ref.relu()
                    rms = ref.pow(2).mean().sqrt().item()
                    ed = (yd[i].float() - ref).pow(2).mean().sqrt().item() / rms
                    ew = (yw[i].float() - ref).pow(2).mean().sqrt().item() / rms
                    mw = (yw[i].float() - ref).abs().max().item() / rms
                    md = (yd[i].float() - ref).abs().max().item() / rms
                    nan = int(torch.isnan(yw[i].float()).sum())
                    stats.append((ed, ew, md, mw, nan))
                ed, ew, md, mw, nan = stats[0]
                model_line = ""
                if B <= 3:
                    # the numpy model's arithmetic (V and U rounded to the 16-bit type, wide accumulation, one output rounding) in torch fp64 on
                    # the device: the kernel's output must be within one 16-bit ulp of it and almost everywhere equal
                    x64 = xs[0].double()
                    ws32 = torch.from_numpy((w * scale[:, None, None, None]).astype(np.float32)).cuda()
                    g0, g1, g2 = ws32[..., 0], ws32[..., 1], ws32[..., 2]
                    U = [g0, 0.5 * ((g0 + g2) + g1), 0.5 * ((g0 + g2) - g1), -g2]
                    U = [u.to(tdt).double() for u in U]
                    P = (W + 1) // 2
                    xp = torch.zeros(B, H + 2, 2 * P + 2, cin, device=dev, dtype=torch.float64)
                    xp[:, 1:H + 1, 1:W + 1] = x64
                    d = [xp[:, :, j:j + 2 * P:2] for j in range(4)]
                    V = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[3] - d[1]]
                    V = [v.float().to(tdt).double() for v in V]
                    M = []
                    for mi in range(4):
                        acc = 0
                        for dh in range(3):
                            acc = acc + torch.einsum("bhpc,oc->bhpo", V[mi][:, dh:dh + H], U[mi][:, :, dh])
                        M.append(acc)
                    ym = torch.zeros(B, H, 2 * P, 128, device=dev, dtype=torch.float64)
                    ym[:, :, 0::2] = M[0] + M[1] + M[2]
                    ym[:, :, 1::2] = M[1] - M[2] - M[3]
                    ym = ym[:, :, :W] + T.double()
                    if flags & R.RD_ADD:
                        ym = ym + rs[0].double()
                    ym = ym.relu()
                    ymr = ym.float().to(tdt).double()
                    yk = yw[0].double()
                    ulp = 2.0 ** (torch.floor(torch.log2(torch.clamp(ym.abs(), min=0.05))) - (7 if dt == R.RD_BF16 else 10))
                    nbad = int(((yk - ymr).abs() > 1.001 * ulp).sum())
                    neq = float((yk == ymr).double().mean())
                    model_line = "   vs the fp64 model of the same arithmetic: %d beyond one ulp, %.4f equal" % (nbad, neq)
                    assert nbad == 0 and neq > 0.97, model_line
                line = "dt %d B %d H %d W %-5d cin %-3d flags %d: rel rms err vs fp32 conv: direct %.3e wino %.3e (x%.2f)  max/rms: %.3e / %.3e  nan %d" % (
                    dt, B, H, W, cin, flags, ed, ew, ew / ed, md, mw, nan)
                line += model_line
                if bench and B == 8:
                    res = {}
                    for name, fn in (("direct", direct), ("wino", wino), ("direct2", direct), ("wino2", wino)):
                        for i in range(3):
                            fn(i % NB)
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        n = 30
                        e0.record()
                        for i in range(n):
                            fn(i % NB)
                        e1.record()
                        torch.cuda.synchronize()
                        res[name] = e0.elapsed_time(e1) * 1e3 / n
                    line += "   us: " + "  ".join("%s %.1f" % kv for kv in res.items())
                print(line, flush=True)
                if not os.environ.get("WINO_DBG"):
                    assert nan == 0 and ew < 4 * ed + 1e-3 and mw < 0.2, "Winograd result out of tolerance"
