"""Fused BasicBlock kernel (csrc/k_block.h) through its development harness tools/micro/block_dev.hip.
    python tools/micro/block_dev.py emu            parity on the CPU emulator (small shapes) against the two unfused launches
    python tools/micro/block_dev.py gpu [bench]    the same on the GPU at full size (bit-equality), and timing
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rangedet_amd import lib as R  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "emu"
c_int, c_void_p = ctypes.c_int, ctypes.c_void_p


def bind(path):
    m = ctypes.CDLL(path)
    m.rdm_block64_packed_bytes.restype = ctypes.c_size_t
    m.rdm_block64_packed_bytes.argtypes = [c_int]
    m.rdm_pack_block64_host.argtypes = [c_void_p] * 4 + [c_int, c_int, c_void_p]
    m.rdm_block64.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    m.rdm_block64_tall.argtypes = [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    m.rdm_last_error.restype = ctypes.c_char_p
    return m


def weights(seed):
    rng = np.random.default_rng(seed)
    w1 = (rng.standard_normal((64, 64, 3, 3)) * 0.06).astype(np.float32)
    w2 = (rng.standard_normal((64, 64, 3, 3)) * 0.06).astype(np.float32)
    s1 = rng.uniform(0.5, 1.5, 64).astype(np.float32)
    s2 = rng.uniform(0.5, 1.5, 64).astype(np.float32)
    t1 = (rng.standard_normal(64) * 0.2).astype(np.float32)
    t2 = (rng.standard_normal(64) * 0.2).astype(np.float32)
    return w1, s1, t1, w2, s2, t2


def pack_block(m, w1, s1, w2, s2, dt):
    out = np.zeros(m.rdm_block64_packed_bytes(w1.shape[1]), np.uint8)
    m.rdm_pack_block64_host(w1.ctypes.data, s1.ctypes.data, w2.ctypes.data, s2.ctypes.data, w1.shape[1], dt, out.ctypes.data)
    return out


if mode == "emu":
    from emu_util import emu_lib, NumpyAllocator, f32_to_bf16_bits
    so = "/tmp/libblock_emu.so"
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(ROOT, "rangedet_amd/csrc/k_block.h")):
        subprocess.check_call("/opt/rocm/lib/llvm/bin/clang++ -x c++ -std=c++17 -O2 -fPIC -shared -DRD_BUILD_NUM_CUS=4 -DRD_BUILD_F16_PRODUCTION_FORMS_ONLY "
                              "-Itests/emu -Iinclude tools/micro/block_dev.hip -o %s 2>/dev/null" % so, shell=True, cwd=ROOT)
    m, L = bind(so), emu_lib()
    # the network's first block: 8 input channels in a 16-channel buffer, projection shortcut
    for (B, H, W, dt, cin) in ((2, 16, 72, R.RD_BF16, 8), (3, 24, 64, R.RD_F16, 5)):
        rng = np.random.default_rng(H + W)
        w1, s1, t1, w2, s2, t2 = weights(B + H)
        w1 = (rng.standard_normal((64, cin, 3, 3)) * 0.1).astype(np.float32)
        wsc = (rng.standard_normal((64, cin)) * 0.3).astype(np.float32)
        ss = rng.uniform(0.5, 1.5, 64).astype(np.float32)
        xf = np.zeros((B, H, W, 16), np.float32)
        xf[..., :cin] = rng.standard_normal((B, H, W, cin))
        xb = f32_to_bf16_bits(xf) if dt == R.RD_BF16 else xf.astype(np.float16).view(np.uint16)
        p1 = L.pack_conv3x3_ex(w1, 1, 16, fold_scale=s1, dtype=dt)
        p2 = L.pack_conv3x3_ex(w2, 1, 64, fold_scale=s2, dtype=dt)
        psc = L.pack_conv1x1_sc(wsc, fold_scale=ss, dtype=dt)
        t = np.zeros((B, H, W, 64), np.uint16)
        yr = np.zeros((B, H, W, 64), np.uint16)
        L.call("rd_conv3x3_bn_act_ex", xb.ctypes.data, 16, 0, p1.ctypes.data, None, t1.ctypes.data, None, 0, 0, None, 0, 0, 0, None,
               t.ctypes.data, 64, 0, B, H, W, cin, 64, 1, R.RD_RELU_POST | R.RD_SCALE_FOLDED, dt, None)
        L.call("rd_conv3x3_bn_act_ex", t.ctypes.data, 64, 0, p2.ctypes.data, None, t2.ctypes.data, None, 0, 0, xb.ctypes.data, 16, 0, cin, psc.ctypes.data,
               yr.ctypes.data, 64, 0, B, H, W, 64, 64, 1, R.RD_ADD | R.RD_RELU_POST | R.RD_SCALE_FOLDED, dt, None)
        pk = pack_block(m, w1, s1, w2, s2, dt)
        y = np.full((B, H, W, 64), 0x7fc0, np.uint16)
        rc = m.rdm_block64(xb.ctypes.data, 16, 0, cin, pk.ctypes.data, t1.ctypes.data, t2.ctypes.data, psc.ctypes.data, y.ctypes.data, 64, 0, B, H, W, dt, None)
        assert rc == 0, m.rdm_last_error()
        bad = y != yr
        print("FIRST block cin %d B %d H %d W %d dt %d: %d of %d values differ from the unfused pair" % (cin, B, H, W, dt, int(bad.sum()), bad.size), flush=True)
        if bad.any():
            idx = np.argwhere(bad)
            print("  first mismatches (b, h, w, c):", idx[:8].tolist())
            sys.exit(1)
    for (B, H, W, dt) in ((2, 16, 72, R.RD_BF16), (1, 11, 33, R.RD_BF16), (1, 8, 100, R.RD_F16), (3, 24, 64, R.RD_BF16)):
        w1, s1, t1, w2, s2, t2 = weights(B + H)
        rng = np.random.default_rng(7 * H + W)
        xf = rng.standard_normal((B, H, W, 64)).astype(np.float32)
        xb = f32_to_bf16_bits(xf) if dt == R.RD_BF16 else xf.astype(np.float16).view(np.uint16)
        # unfused reference: two production launches
        p1 = L.pack_conv3x3_ex(w1, 1, 64, fold_scale=s1, dtype=dt)
        p2 = L.pack_conv3x3_ex(w2, 1, 64, fold_scale=s2, dtype=dt)
        t = np.zeros((B, H, W, 64), np.uint16)
        yr = np.zeros((B, H, W, 64), np.uint16)
        L.call("rd_conv3x3_bn_act_ex", xb.ctypes.data, 64, 0, p1.ctypes.data, None, t1.ctypes.data, None, 0, 0, None, 0, 0, 0, None,
               t.ctypes.data, 64, 0, B, H, W, 64, 64, 1, R.RD_RELU_POST | R.RD_SCALE_FOLDED, dt, None)
        L.call("rd_conv3x3_bn_act_ex", t.ctypes.data, 64, 0, p2.ctypes.data, None, t2.ctypes.data, xb.ctypes.data, 64, 0, None, 0, 0, 0, None,
               yr.ctypes.data, 64, 0, B, H, W, 64, 64, 1, R.RD_ADD | R.RD_RELU_POST | R.RD_SCALE_FOLDED, dt, None)
        pk = pack_block(m, w1, s1, w2, s2, dt)
        y = np.full((B, H, W, 64), 0x7fc0, np.uint16)
        rc = m.rdm_block64(xb.ctypes.data, 64, 0, 64, pk.ctypes.data, t1.ctypes.data, t2.ctypes.data, None, y.ctypes.data, 64, 0, B, H, W, dt, None)
        assert rc == 0, m.rdm_last_error()
        bad = y != yr
        print("B %d H %d W %d dt %d: %d of %d values differ from the unfused pair" % (B, H, W, dt, int(bad.sum()), bad.size), flush=True)
        if dt == R.RD_BF16:     # the 16 x 32-tile experiment (tools/micro/k_block16.h)
            y2 = np.full((B, H, W, 64), 0x7fc0, np.uint16)
            rc = m.rdm_block64_tall(xb.ctypes.data, 64, 0, pk.ctypes.data, t1.ctypes.data, t2.ctypes.data, y2.ctypes.data, 64, 0, B, H, W, dt, None)
            assert rc == 0, m.rdm_last_error()
            bad2 = y2 != yr
            print("   tall tiles: %d of %d values differ" % (int(bad2.sum()), bad2.size), flush=True)
            if bad2.any():
                idx = np.argwhere(bad2)
                print("  first mismatches (b, h, w, c):", idx[:8].tolist())
                print("  rows:", sorted(set(idx[:, 1].tolist()))[:24], "cols:", sorted(set(idx[:, 2].tolist()))[:40])
                sys.exit(1)
        if bad.any():
            idx = np.argwhere(bad)
            print("  first mismatches (b, h, w, c):", idx[:8].tolist())
            print("  rows with mismatches:", sorted(set(idx[:, 1].tolist()))[:20], "cols:", sorted(set(idx[:, 2].tolist()))[:40])
            sys.exit(1)
    print("emu parity ok")
else:
    import torch
    so = os.environ.get("BLOCK_SO", os.path.join(ROOT, "tools/micro/libblock_dev.so"))
    m, L = bind(so), R.get_lib()
    st = torch.cuda.current_stream().cuda_stream
    dev = "cuda"
    for dt, tdt in ((R.RD_BF16, torch.bfloat16), (R.RD_F16, torch.float16)):
        for (B, H, W) in ((8, 64, 2656), (8, 64, 1328), (3, 64, 2650), (1, 20, 77)):
            w1, s1, t1, w2, s2, t2 = weights(B + W)
            NB = 3
            xs = [torch.randn(B, H, W, 64, device=dev).relu().to(tdt) for _ in range(NB)]
            p1 = torch.from_numpy(L.pack_conv3x3_ex(w1, 1, 64, fold_scale=s1, dtype=dt)).cuda()
            p2 = torch.from_numpy(L.pack_conv3x3_ex(w2, 1, 64, fold_scale=s2, dtype=dt)).cuda()
            pk = torch.from_numpy(pack_block(m, w1, s1, w2, s2, dt)).cuda()
            T1, T2 = torch.from_numpy(t1).cuda(), torch.from_numpy(t2).cuda()
            ts = [torch.empty(B, H, W, 64, device=dev, dtype=tdt) for _ in range(NB)]
            yr = [torch.empty(B, H, W, 64, device=dev, dtype=tdt) for _ in range(NB)]
            yf = [torch.full((B, H, W, 64), float("nan"), device=dev, dtype=tdt) for _ in range(NB)]

            def unfused(i):
                L.call("rd_conv3x3_bn_act_ex", xs[i].data_ptr(), 64, 0, p1.data_ptr(), None, T1.data_ptr(), None, 0, 0, None, 0, 0, 0, None,
                       ts[i].data_ptr(), 64, 0, B, H, W, 64, 64, 1, R.RD_RELU_POST | R.RD_SCALE_FOLDED, dt, st)
                L.call("rd_conv3x3_bn_act_ex", ts[i].data_ptr(), 64, 0, p2.data_ptr(), None, T2.data_ptr(), xs[i].data_ptr(), 64, 0, None, 0, 0, 0, None,
                       yr[i].data_ptr(), 64, 0, B, H, W, 64, 64, 1, R.RD_ADD | R.RD_RELU_POST | R.RD_SCALE_FOLDED, dt, st)

            def fused(i):
                rc = m.rdm_block64(xs[i].data_ptr(), 64, 0, 64, pk.data_ptr(), T1.data_ptr(), T2.data_ptr(), None, yf[i].data_ptr(), 64, 0, B, H, W, dt, st)
                assert rc == 0, m.rdm_last_error()
            for i in range(NB):
                unfused(i)
                fused(i)
            torch.cuda.synchronize()
            nbad = sum(int((yf[i].view(torch.int16) != yr[i].view(torch.int16)).sum()) for i in range(NB))
            line = "dt %d B %d H %d W %-5d: %d values differ from the unfused pair" % (dt, B, H, W, nbad)
            if len(sys.argv) > 2 and B == 8:
                res = {}
                ytl = [torch.full((B, H, W, 64), float("nan"), device=dev, dtype=tdt) for _ in range(NB)]

                def tall(i):
                    rc = m.rdm_block64_tall(xs[i].data_ptr(), 64, 0, pk.data_ptr(), T1.data_ptr(), T2.data_ptr(), ytl[i].data_ptr(), 64, 0, B, H, W, dt, st)
                    assert rc == 0, m.rdm_last_error()
                variants = [("unfused", unfused), ("fused", fused)]
                if dt == R.RD_BF16:
                    for i in range(NB):
                        tall(i)
                    torch.cuda.synchronize()
                    nb2 = sum(int((ytl[i].view(torch.int16) != yr[i].view(torch.int16)).sum()) for i in range(NB))
                    line += "  [tall: %d differ]" % nb2
                    variants.append(("tall", tall))
                for name, fn in variants + [(n + "2", f) for n, f in variants]:
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
                line += "   us per block: " + "  ".join("%s %.1f" % kv for kv in res.items())
            print(line, flush=True)
            assert nbad == 0
