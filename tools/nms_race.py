"""Does the batched weighted-NMS chain give the same result when other kernels share the GPU?
Round 5: bench.py's per-step digests differed from run to run with more than one batch in flight -- same candidates, keep counts off by one
or two.  This tool found the cause: the chain (rd_wnms_4c_batched on one pipeline's filtered detections) is replayed REPS
times on a side stream while another pipeline's forward runs on a second stream, and compared with the result of the same call on an
idle GPU; on a difference the intermediate arrays in the workspace (prep records, processing order, thr / vote matrices, alive list)
are compared stage by stage.
    [TIE=stable] [LOAD=forward|none|matmul|kind:<plan step kind>|sweep] [LREP=3] [RD_WNMS_...=1] python tools/nms_race.py [reps]
  LOAD=sweep   every plan step on its own as the concurrent load (which launches matter: the cout-64 ones, two workgroups per CU)
  RANGEDET_HIP_LIB=<a build with the SLP vectoriser>  reproduces the fault; the shipped build gives 0 of N
(Round 6: the RD_* variables named here are DEVELOPMENT switches -- the release library ignores them.  Build the A/B library with
`python -m rangedet_amd.build --dev` and run with RANGEDET_HIP_LIB=rangedet_amd/librangedet_hip_dev.so RD_DEV_SWITCHES=1; tools/exp/ab.sh does both.)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rangedet_amd import lib as rdlib, synth  # noqa: E402
from rangedet_amd.pipeline import RangeDetPipeline  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = 8
P = synth.make_weights(seed=18)
pa = RangeDetPipeline(P, dtype=rdlib.RD_BF16, batch=B, wnms_cap=8192, tie_order=os.environ.get("TIE", "reference"))
pb = RangeDetPipeline(P, dtype=rdlib.RD_BF16, batch=B, wnms_cap=8192)
fr = synth.make_batch(list(range(B)), lib=pa.lib, alloc=pa.alloc)
fr2 = synth.make_batch(list(range(B, 2 * B)), lib=pa.lib, alloc=pa.alloc)
A = pa.alloc
pa.enqueue(fr)                       # fills pa's filtered detections (dets / count)
torch.cuda.synchronize()
bp = pa.bpost


def snap():
    torch.cuda.synchronize()
    nk = np.array(A.to_numpy(A.view_i32(bp.nkeep, (B,))))
    keep = np.array(A.to_numpy(A.view_i32(bp.keep, (B, bp.cap))))
    rows = np.array(A.to_numpy(A.view_f32(bp.out, (B, bp.cap, 12))))
    return nk, [keep[b, :nk[b]].copy() for b in range(B)], [rows[b, :nk[b]].copy() for b in range(B)]


def a256(v):
    return (v + 255) & ~255


def stages():
    """Intermediate arrays of the chain, read out of the workspace (layout: rd_api.hip wnms_ws_carve)."""
    torch.cuda.synchronize()
    cap = bp.cap
    nw = (cap + 63) // 64
    per = pa.lib.raw("rd_wnms_workspace_bytes")(cap)
    ws = np.array(A.to_numpy(bp.ws_w))
    base_off = (-bp.ws_w.data_ptr()) % 256
    cnt = np.minimum(np.array(A.to_numpy(A.view_i32(bp.count, (B,)))), cap)
    out = {}
    for b in range(B):
        f = ws[base_off + b * per: base_off + (b + 1) * per]
        K = int(cnt[b])
        o = 0
        prep = f[o:o + cap * 26 * 4].view(np.float32).reshape(cap, 26)[:K]; o += a256(cap * 26 * 4)
        thr = f[o:o + cap * nw * 8].view(np.uint64).reshape(cap, nw)[:K]; o += a256(cap * nw * 8)
        vote = f[o:o + cap * nw * 8].view(np.uint64).reshape(cap, nw)[:K]; o += a256(cap * nw * 8)
        o += a256(cap * nw * 8)
        keep_q = f[o:o + cap * 4].view(np.int32); o += a256(cap * 4)
        order = f[o:o + cap * 4].view(np.int32)[:K]; o += a256(cap * 4)
        alive = f[o:o + cap * 4].view(np.int32); o += a256(cap * 4)
        o += a256(cap * 4)
        o += a256((cap + 2) * 8 + 3 * (cap // 32 + 2) * 4)
        o += a256(nw * 8)
        nalive = int(f[o:o + 4].view(np.int32)[0])
        # the words the scan reads: from each row's diagonal word on, columns < K
        kw = (K + 63) // 64
        rowi = np.arange(K)[:, None] // 64
        m = (np.arange(kw)[None, :] >= rowi)
        out[b] = dict(prep=prep.view(np.uint32).copy(), order=order.copy(), thr=np.where(m, thr[:, :kw], 0), vote=np.where(m, vote[:, :kw], 0),
                      nalive=nalive, alive=alive[:max(nalive, 0)].copy())
    return out


def first_diff(s, ref):
    for b in range(B):
        for k in ("prep", "order", "nalive", "alive", "thr", "vote"):
            x, y = s[b][k], ref[b][k]
            if np.shape(x) != np.shape(y) or not np.array_equal(x, y):
                msg = "frame %d stage %s" % (b, k)
                if k in ("thr", "vote", "prep") and np.shape(x) == np.shape(y):
                    r, c = np.nonzero(x != y)
                    if k == "prep":
                        q = int(r[0])
                        cn = y[q].view(np.float32)[:8].reshape(4, 2).astype(np.float64)
                        host = 0.5 * abs(sum(cn[i, 0] * cn[(i + 1) % 4, 1] - cn[(i + 1) % 4, 0] * cn[i, 1] for i in range(4)))
                        wrong = x[q, 12]
                        where = [(bb, int(np.nonzero(ref[bb]["prep"][:, 12] == wrong)[0][0])) for bb in range(B) if (ref[bb]["prep"][:, 12] == wrong).any()]
                        raw = np.array(A.to_numpy(A.view_f32(bp.dets, (B, bp.k, 12))))[b, q, :12]
                        wv = x[q, 12:13].view(np.float32)[0]
                        hits = [(i, j) for i in range(12) for j in range(12) if i != j and np.float32(raw[i] - raw[j]) == wv]
                        msg += " [wrong value == raw[i]-raw[j] for (i,j) in %s; lane %d]" % (hits, q % 64)
                        msg += " [row %d: shoelace area %.5f, idle %.5f, now %.5f; the wrong bits are the idle area of (frame,row) %s; cols %s]" % (
                            q, host, float(y[q, 12:13].view(np.float32)[0]), float(x[q, 12:13].view(np.float32)[0]), where[:4], sorted(set(c_ for c_ in np.nonzero(x != y)[1].tolist())))
                    msg += ": %d entries, first (row %d, col %d) %x vs %x; rows %s" % (len(r), r[0], c[0], int(x[r[0], c[0]]), int(y[r[0], c[0]]), sorted(set(r.tolist()))[:10])
                return msg
    return "none of the stages"


ONE = os.environ.get("RD_WNMS_ONE_ROUND") == "1"
bp.enqueue_nms()
base = snap()
base_st = stages()
for _ in range(3):                   # idle GPU: reproducible?
    bp.enqueue_nms()
    s = snap()
    assert all(np.array_equal(a, b) for a, b in zip(s[1], base[1])) and all(np.array_equal(a, b) for a, b in zip(s[2], base[2])), "differs on an idle GPU"
print("idle: reproducible; keep counts", base[0].tolist())
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
LOAD = os.environ.get("LOAD", "forward")
MA = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
MB = torch.empty_like(MA)
DEV = {}


def subkind(b):
    k = b["kind"]
    if k == "conv":
        return "conv_head" if b.get("head") else "conv_cat" if b.get("x2") is not None else "conv_ex" if b.get("ex") else "conv_plain"
    return k


LIDX = [i for i, b in enumerate(pb.exe._bound) if LOAD == "kind:" + subkind(b)]
LREP = int(os.environ.get("LREP", "3"))
if LOAD.startswith("kind:"):
    print("steps of", LOAD, ":", len(LIDX), "of kinds", sorted(set(subkind(b) for b in pb.exe._bound)))
    pb.exe.forward(fr2)
    torch.cuda.synchronize()
# a second victim: a plain element-wise kernel of the library (no LDS, no atomics)
NV = 1 << 21
VY, VX = torch.randn(NV, device="cuda"), torch.randn(NV, device="cuda")
VO = torch.empty(NV, device="cuda")
pa.lib.call("rd_edge_atan2f", VY.data_ptr(), VX.data_ptr(), NV, VO.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
VBASE = VO.clone()
vbad = 0
if LOAD == "sweep":       # every plan step on its own as the concurrent load
    pb.exe.forward(fr2)
    torch.cuda.synchronize()
    for i, b in enumerate(pb.exe._bound):
        nb = 0
        for r in range(reps):
            with torch.cuda.stream(s1):
                for _ in range(LREP):
                    pb.exe.forward(fr2, only=i, dev=DEV)
            with torch.cuda.stream(s2):
                bp.enqueue_nms(stream=s2)
            s = snap()
            nb += not (all(np.array_equal(a, c) for a, c in zip(s[1], base[1])) and all(np.array_equal(a, c) for a, c in zip(s[2], base[2])))
        x = b.get("x")
        print("step %2d %-10s %-28s cin %s cout %s W %s: %d of %d differ" % (i, subkind(b), b.get("name", "")[:28], b.get("cin"), b.get("cout"), getattr(x, "W", None), nb, reps), flush=True)
    sys.exit(0)
bad = 0
for r in range(reps):
    with torch.cuda.stream(s1):
        if LOAD == "matmul":
            for _ in range(20):
                torch.mm(MA, MA, out=MB)
        elif LOAD == "forward":
            pb.exe.forward(fr2)          # conv kernels of another batch
        elif LOAD.startswith("kind:"):
            for _ in range(LREP):
                for i in LIDX:
                    pb.exe.forward(fr2, only=i, dev=DEV)
    with torch.cuda.stream(s2):
        bp.enqueue_nms(stream=s2)
        pa.lib.call("rd_edge_atan2f", VY.data_ptr(), VX.data_ptr(), NV, VO.data_ptr(), s2.cuda_stream)
    s = snap()
    nvd = int((VO.view(torch.int32) != VBASE.view(torch.int32)).sum())
    if nvd:
        vbad += 1
        if vbad <= 3:
            ii = torch.nonzero(VO.view(torch.int32) != VBASE.view(torch.int32)).flatten()
            print("   atan2f victim: %d elements differ, first indices %s" % (nvd, ii[:20].tolist()))
    same = all(np.array_equal(a, b) for a, b in zip(s[1], base[1])) and all(np.array_equal(a, b) for a, b in zip(s[2], base[2]))
    if not same:
        bad += 1
        if bad <= 6:
            print("   first differing stage:", first_diff(stages(), base_st))
            print("rep %d differs: keep counts %s (idle %s); first differing frame %d" % (
                r, s[0].tolist(), base[0].tolist(), next(b for b in range(B) if not (np.array_equal(s[1][b], base[1][b]) and np.array_equal(s[2][b], base[2][b])))))
print("%s: %d of %d concurrent replays differ from the idle result (element-wise victim: %d)" % (" ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("RD_") or k in ("TIE", "LOAD")) or "default", bad, reps, vbad))
