"""Per-plan-step timing on the GPU (each step replayed `reps` times back to back, torch events): a tuning aid.
    python tools/profile_steps.py [bf16|f32] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rangedet_amd import lib as rdlib, synth  # noqa: E402
from rangedet_amd.pipeline import RangeDetPipeline  # noqa: E402

dt = {"f32": rdlib.RD_F32, "f16": rdlib.RD_F16}.get(sys.argv[1] if len(sys.argv) > 1 else "bf16", rdlib.RD_BF16)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pipe = RangeDetPipeline(synth.make_weights(seed=18), dtype=dt, batch=B)
fr = synth.make_batch(list(range(B)))
pipe.enqueue(fr)
torch.cuda.synchronize()
ex = pipe.exe
esz = 2 if dt in rdlib.H16 else 4
tot = 0.0
rows = []
for i, st in enumerate(pipe.plan.steps):
    dev = {}
    ex.forward(fr, only=i, dev=dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ex.forward(fr, only=i, dev=dev)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tot += us
    k = st["kind"]
    fl = 0.0
    by = 0.0          # bytes THIS launch form has to move (fused form: input once incl. a stride-2 conv's skipped columns, output once,
    desc = ""         # residual / shortcut input once) -- against the HBM roof next to the FLOPs against the MFMA roof
    if k == "conv_pair":   # two tower convs of one shape in one launch
        a = st["a"]
        o = a["out"]
        fl = 2 * B * 2.0 * o.H * o.W * a["cin"] * a["cout"] * 9
        by = 2 * B * esz * o.H * o.W * (a["cin"] + (0 if a.get("head") else a["cout"]))
        desc = "%s %d->%d x2 W%d" % (st["name"].replace("rpn_", "").replace("_conv", ""), a["cin"], a["cout"], o.W)
    elif k == "block":     # a fused 64-channel BasicBlock: the FLOPs of its two convs (+ the 1x1 shortcut)
        o = st["out"]
        c1 = st["a"]["cin"]
        fl = B * 2.0 * o.H * o.W * 64 * (9 * c1 + 9 * 64 + (c1 if st["b"].get("sc") else 0))
        by = B * esz * o.H * o.W * (st["x"].cs - st["x"].co if c1 <= 16 else c1) + B * esz * o.H * o.W * 64
        desc = "%s %d->64->64 W%d%s" % (st["name"].replace("_conv1 + ", " + ").split(" + ")[0] + " block", c1, o.W, " +sc" if st["b"].get("sc") else "")
    elif k in ("conv", "deconv"):
        o = st["out"]
        fl = B * 2.0 * o.H * o.W * st["cin"] * st["cout"] * st["k"][0] * st["k"][1] / (st["stride_w"] if k == "deconv" else 1)
        x = st["x"]
        by = B * esz * (x.H * x.W * st["cin"] + (0 if st.get("head") else o.H * o.W * st["cout"]))
        if st.get("res") is not None:
            by += B * esz * o.H * o.W * st["cout"]
        if st.get("sc"):       # the fused 1x1 projection shortcut reads the block input at the conv's stride: every 128-byte line of it is touched
            sx = st["sc_x"]
            by += B * esz * sx.H * sx.W * st["sc"]["cin"]
        if st.get("x2") is not None:
            by += B * esz * x.H * x.W * (st["x2"].cs - st["x2"].co)
        desc = "%s %d->%d k%s W%d->%d s%d" % (st["name"], st["cin"], st["cout"], st["k"], st["x"].W, o.W, st["stride_w"])
    elif k == "meta":
        fl = B * 19.29e9
        by = B * 64 * 2656 * (128 * esz + 12)
        desc = "meta unit"
    else:
        desc = st.get("name", "")
    if st.get("m16"):
        desc += " [16x16x32]"
    rows.append((us, k, desc, fl))
    # tiles per resident workgroup slot of the persistent 3x3 kernel (the rule of launch_conv3, k_conv3.h, restated for this
    # report: 8 x 32 tiles (RD_CONV_WIDE=0: 8 x 30) with two workgroups per CU, except a fused output conv wider than 1400 columns that is not
    # in the 16 x 16 x 32 form (step key "m16"; round 6): 8 x 62, one per CU;
    # a stride-2 conv runs on the pixel-pair view = its output grid, a transposed conv phase on its input grid)
    tps = ""
    if dt in rdlib.H16 and k == "block":
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        ntiles = -(-st["out"].W // 32) * -(-st["out"].H // 8) * B
        tps = "  %5d tiles / %d slots = %5.2f" % (ntiles, 2 * cus, ntiles / (2 * cus))
    elif dt in rdlib.H16 and k == "conv_pair":
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        ntiles = 2 * -(-st["out"].W // 32) * -(-st["out"].H // 8) * B
        tps = "  %5d tiles / %d slots = %5.2f" % (ntiles, 2 * cus, ntiles / (2 * cus))
    elif dt in rdlib.H16 and k in ("conv", "deconv") and st["k"][0] == 3:
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        Wt = st["x"].W if k == "deconv" else st["out"].W
        wide_head = bool(st.get("head")) and Wt > 1400 and not st.get("m16")
        tw, slots = (62, cus) if wide_head else (32 if os.environ.get("RD_CONV_WIDE", "1") != "0" else 30, 2 * cus)
        ntiles = -(-Wt // tw) * -(-st["out"].H // 8) * B
        tps = "  %5d tiles / %d slots = %5.2f" % (ntiles, slots, ntiles / slots) + (" per phase" if k == "deconv" else "")
    tf, tbs = (fl / us / 1e6 if fl else 0), (by / us / 1e6 if by else 0)
    roof = ""
    if fl or by:
        fm, fh = tf / 2500.0, tbs / 8.0
        roof = "  %5.2f TB/s  %4.1f %% of the %s roof" % (tbs, 100 * max(fm, fh), "MFMA" if fm >= fh else "HBM")
    print("%3d %-9s %-52s %8.1f us %8.1f TFLOP/s%s%s" % (i, k, desc[:52], us, tf, roof, tps), flush=True)
print("sum of steps: %.1f us" % tot)
