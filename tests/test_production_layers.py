"""Tight on-hardware parity of EVERY production 16-bit launch at the production geometry.

The unit tests of test_kernels.py run the persistent 3x3 kernel / the Meta-Kernel at sizes where every workgroup owns one
tile.  This test runs the real lowered plan of rangedet_veh_wo_aug_4_18e at B = 8, 64 x 2656 -- 5 312 tiles on 512 resident
slots at full width, the 83rd column tile, the W = 1328 / 664 / 332 layers, the 0.75-tile-per-slot W = 166 launches, the
stride-2 pixel-pair views, the fused projection shortcuts, every transposed conv with all its phases in one launch, the level-0
tower convs that read the never-materialised concat of the agg3 feature map and the range image from two tensors, the
fused tower output convs on both tile shapes and the Meta-Kernel's register prefetch of the next tile -- ONE PLAN STEP AT A
TIME: the step's actual device inputs are read back (exact: they are 16-bit values), the layer is recomputed by PyTorch-CPU in
fp32 from those inputs and the SAME 16-bit weights the packer makes (folded BatchNorm scale, dla_backbone.py:18-56,117-127;
mxnext/complicate.py:26-45), and the device output must agree PER ELEMENT to one rounding of the output type
(|got - ref| <= 2^-8 |ref| for bf16, 2^-11 for fp16, plus fp32 summation-order noise).  A fused BasicBlock (round 5,
rd_block64_bn_act) must be BIT-IDENTICAL to the two launches it replaces, which are replayed on scratch buffers and checked per
element like every other conv (conv2 on the device's own intermediate).  The Meta-Kernel step is checked
against oracle/graph_ref.meta_kernel_unit (meta_kernel.py:166-240) with test_meta_kernel_unit's error model.
"""
import time

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import HIP_ONLY
from oracle import graph_ref as G
from oracle import input_ref as IR
from rangedet_amd import lib as R
from rangedet_amd import synth
from rangedet_amd.lower import TRef
from rangedet_amd.runtime import bn_affine

BF16, F16 = R.RD_BF16, R.RD_F16


def _round_t(t, dt):
    """torch float32 -> rounded to the 16-bit type -> float32 (multi-threaded; numpy's astype is single-threaded)."""
    return t.to(torch.bfloat16 if dt == BF16 else torch.float16).to(torch.float32)


def _expand(w, cmap):
    """logical input channels -> the physical channels of a concat buffer with alignment padding (runtime.Executor._bind)."""
    if not cmap:
        return w
    wp = np.zeros((w.shape[0], len(cmap)) + w.shape[2:], np.float32)
    for pc, lc in enumerate(cmap):
        if lc >= 0:
            wp[:, pc] = w[:, lc]
    return wp


class _Host:
    """Host copies (NCHW float32 torch tensors) of the plan's activation buffers, keyed by logical buffer: every TRef is
    written by the steps that name it as `out`, so one download per produced tensor instead of one per use."""

    def __init__(self, exe, plan):
        self.exe, self.cache = exe, {}
        self.last = {}
        for i, st in enumerate(plan.steps):
            for v in st.values():
                if isinstance(v, TRef):
                    self.last[v.buf] = i

    def get(self, ref, fresh=False):
        key = (ref.buf, ref.co, ref.C)
        if fresh or key not in self.cache:
            self.cache[key] = torch.from_numpy(np.ascontiguousarray(self.exe.debug_tensor(ref)))
        return self.cache[key]

    def retire(self, i):
        for key in [k for k in self.cache if self.last.get(k[0], -1) <= i]:
            del self.cache[key]

    def invalidate(self, ref):
        for key in [k for k in self.cache if k[0] == ref.buf]:
            del self.cache[key]


def _meta_ref_and_sigma(data, coord, P, name, u):
    """oracle/graph_ref.meta_kernel_unit (meta_kernel.py:166-240 + dla_backbone.py:92-97) for ONE image, restated here only to
    also return the per-element standard deviation the 16-bit Meta-Kernel's roundings imply (u = half an ulp of the type):
      h (32 hidden units) and s1*W1 are rounded            -> var(w_c)  = u^2/3 * 2 * sum_j (W1_cj h_j)^2
      a = relu(s1 d w + t1) is rounded                      -> var(a)    = (s1 d)^2 var(w_c) + u^2/3 a^2
      A (576 -> 64) is rounded                              -> var(pre)  = sum A^2 var(a) + u^2/3 sum (A a)^2
      y = relu(s2 pre + t2)                                 -> sigma_y   = |s2| sqrt(var(pre))     (the output rounding is added by the caller)
    The relative coordinates / the 3 -> 32 layer are fp32-accurate on the device (hi + lo split operands)."""
    B, C, H, W = data.shape
    pre_, Wn = name + "_", str(W)
    T = G.T
    cs = F.unfold(coord, 3, padding=1).view(B, 3, 9, H, W)
    rel = (cs - coord.unsqueeze(2)).reshape(B, 3, 9 * H, W)
    h = F.relu(F.conv2d(rel, T(P[pre_ + Wn + "_mlp0_weight"]), T(P[pre_ + Wn + "_mlp0_bias"])))
    W1 = T(P[pre_ + Wn + "_mlp1_weight"])
    wts = F.conv2d(h, W1, T(P[pre_ + Wn + "_mlp1_bias"])).view(B, 64, 9, H, W)
    t1q = F.conv2d(h * h, W1 * W1).view(B, 64, 9, H, W)                  # sum_j (W1_cj h_j)^2
    ds = F.unfold(data, 3, padding=1).view(B, C, 9, H, W)
    s1, t1 = (torch.from_numpy(v) for v in bn_affine(P, name + "point_wise_mlp_bn1", G.EPS))
    s2, t2 = (torch.from_numpy(v) for v in bn_affine(P, name + "aggregation_bn1", G.EPS))
    a = F.relu((ds * wts).reshape(B, C * 9, H, W) * s1.view(1, -1, 1, 1) + t1.view(1, -1, 1, 1))
    q = u * u / 3.0
    var_a = (ds.reshape(B, C * 9, H, W) * s1.view(1, -1, 1, 1)) ** 2 * (2.0 * q) * t1q.reshape(B, C * 9, H, W) + 2.0 * q * a * a
    A = T(P[name + "aggregation_conv1_weight"])
    pre = F.conv2d(a, A)
    var = F.conv2d(var_a, A * A)                                        # (the A-rounding term is the second q a^2 above)
    y = F.relu(pre * s2.view(1, -1, 1, 1) + t2.view(1, -1, 1, 1))
    return y, s2.abs().view(1, -1, 1, 1) * var.sqrt()


def _check(name, got, ref, dt, report, extra_abs=0.0):
    """per-element: one rounding of the output type (half an ulp <= 2^-8 |v| for bf16's 8 significant bits, 2^-11 |v| for
    fp16's 11) + fp32 summation-order noise (1e-5 of the largest value; the BatchNorm shift enters the accumulators as a
    16-bit hi + lo pair: 2^-17 of the shift in bf16 mode, passed as extra_abs)"""
    u = 2.0 ** -8 if dt == BF16 else 2.0 ** -11
    scale = float(ref.abs().max())
    tol = u * ref.abs() + (1e-5 * max(1.0, scale) + extra_abs)
    err = (got - ref).abs()
    bad = err > tol
    nbad = int(bad.sum())
    worst = float((err / tol).max())
    report.append((name, tuple(ref.shape), worst, nbad))
    return nbad == 0


@pytest.mark.slow
@pytest.mark.parametrize("be", HIP_ONLY, indirect=True)
@pytest.mark.parametrize("dt,variant", [(BF16, "veh"), (F16, "veh"), (F16, "kitti")], ids=["bf16", "f16", "kitti-f16"])
def test_every_production_launch_tight_at_full_geometry(be, dt, variant):
    """variant "veh": rangedet_veh_wo_aug_4_18e at 64 x 2656 (BASELINE configs[1]); "kitti": the two-class configuration of BASELINE
    configs[4] at 64 x 2048 x 5 in fp16 -- W = 2048 / 1024 / 512 / 256 / 128 tile lists, a 5-channel first layer, 128-channel tower
    outputs written for the separate two-class output convs."""
    from rangedet_amd.pipeline import RangeDetPipeline
    from rangedet_amd.config import rangedet_veh_wo_aug_4_18e as cfgmod
    torch.set_num_threads(min(64, torch.get_num_threads()))
    t00 = time.time()
    B = 8
    if variant == "kitti":
        Wk = 2048
        P = synth.make_weights(seed=18, width=Wk, in_ch=cfgmod.KITTI_INPUT_CHANNELS, num_classes=2)
        pipe = RangeDetPipeline(P, dtype=dt, batch=B, wnms_cap=4096, lib=be.lib, alloc=be.alloc, variant="kitti", feat_size=(64, Wk),
                                pad_field=(64, Wk), pre_nms_top_n={'veh': 50000, 'ped': 5000})
        fr = IR.make_batch(list(range(B)), W=Wk, pad_W=Wk, H=64)
        fr['input_data'] = np.ascontiguousarray(fr['input_data'][:, [0, 3, 4, 5, 1]])     # KITTI channel order: range, x, y, z, intensity
    else:
        P = synth.make_weights(seed=18)
        pipe = RangeDetPipeline(P, dtype=dt, batch=B, wnms_cap=4096, lib=be.lib, alloc=be.alloc)
        fr = IR.make_batch(list(range(B)))
    plan, exe = pipe.plan, pipe.exe
    dev = {}
    host = _Host(exe, plan)
    report, failed = [], []
    seen_forms = set()
    for i, st in enumerate(plan.steps):
        k = st["kind"]
        if k == "block":
            # A fused BasicBlock (rd_block64_bn_act, round 5).  Its bound is "two roundings" only in the sense that the intermediate tensor is
            # rounded like the stored one: the launch must be BIT-IDENTICAL to the two launches it replaces, and those two are checked per
            # element to one output rounding each -- conv1 against torch on the block input, conv2 (+ shortcut) on the DEVICE's own
            # intermediate -- so no tolerance is widened for the fused form.
            a_, b_ = st["a"], st["b"]
            xr, orf = st["x"], st["out"]
            x = host.get(xr)
            exe.forward(fr, only=i, dev=dev)
            host.invalidate(orf)
            got = host.get(orf, fresh=True)
            L_, A_ = be.lib, be.alloc
            (s1, t1), (s2, t2) = bn_affine(P, a_["bn"], a_["eps"]), bn_affine(P, b_["bn"], b_["eps"])
            w1, w2 = (np.asarray(P[c_["name"] + "_weight"], np.float32) for c_ in (a_, b_))
            w1 = _expand(w1, a_.get("cmap"))
            cin1 = w1.shape[1]
            if x.shape[1] < cin1:     # (the first block: the buffer's zero channels past the real ones)
                x = torch.cat([x, torch.zeros(x.shape[0], cin1 - x.shape[1], x.shape[2], x.shape[3])], 1)
            p1 = A_.upload(L_.pack_conv3x3_ex(w1, 1, xr.cs, fold_scale=s1, dtype=dt))
            p2 = A_.upload(L_.pack_conv3x3_ex(w2, 1, 64, fold_scale=s2, dtype=dt))
            shift2 = t2.astype(np.float64)
            psc = None
            if b_.get("sc"):
                sc = b_["sc"]
                wsc = _expand(np.asarray(P[sc["name"] + "_weight"], np.float32).reshape(64, -1), sc.get("cmap"))
                if wsc.shape[1] < cin1:
                    wsc = np.concatenate([wsc, np.zeros((64, cin1 - wsc.shape[1]), np.float32)], 1)
                ss, ts = bn_affine(P, sc["bn"], sc["eps"])
                psc = A_.upload(L_.pack_conv1x1_sc(wsc, fold_scale=ss, dtype=dt))
                shift2 = shift2 + ts
            d1, d2 = A_.upload(t1), A_.upload(shift2.astype(np.float32))
            nby = B * xr.H * xr.W * 64 * 2
            exe._phys[-1], exe._phys[-2] = A_.alloc(nby), A_.alloc(nby)
            tr, yr = TRef(-1, 64, xr.H, xr.W, 64, 0), TRef(-2, 64, xr.H, xr.W, 64, 0)
            FO = R.RD_SCALE_FOLDED
            L_.call("rd_conv3x3_bn_act_ex", exe.p(xr), xr.cs, xr.co, A_.ptr(p1), None, A_.ptr(d1), None, 0, 0, None, 0, 0, 0, None, exe.p(tr), 64, 0,
                    B, xr.H, xr.W, cin1, 64, 1, R.RD_RELU_POST | FO, dt, A_.stream)
            L_.call("rd_conv3x3_bn_act_ex", exe.p(tr), 64, 0, A_.ptr(p2), None, A_.ptr(d2), None if psc is not None else exe.p(xr),
                    0 if psc is not None else xr.cs, 0 if psc is not None else xr.co, exe.p(xr) if psc is not None else None,
                    xr.cs if psc is not None else 0, xr.co if psc is not None else 0, cin1 if psc is not None else 0,
                    A_.ptr(psc) if psc is not None else None, exe.p(yr), 64, 0, B, xr.H, xr.W, 64, 64, 1, R.RD_ADD | R.RD_RELU_POST | FO, dt, A_.stream)
            t_dev = torch.from_numpy(np.ascontiguousarray(exe.debug_tensor(tr)))
            y_two = torch.from_numpy(np.ascontiguousarray(exe.debug_tensor(yr)))
            del exe._phys[-1], exe._phys[-2]
            nd = int((got != y_two).sum())
            report.append((st["name"] + " [fused block vs its two launches, differing elements]", tuple(got.shape), 0.0, nd))
            if nd:
                failed.append(st["name"] + " (fused block differs from the two launches on %d elements)" % nd)
            w1q = _round_t(torch.from_numpy(w1 * s1[:, None, None, None]), dt)
            w2q = _round_t(torch.from_numpy(w2 * s2[:, None, None, None]), dt)
            r1 = torch.relu(F.conv2d(x, w1q, padding=1) + torch.from_numpy(t1)[None, :, None, None])
            se1 = 2.0 ** -16 * float(np.abs(t1).max()) if dt == BF16 else 0.0
            if not _check(a_["name"], t_dev, r1, dt, report, extra_abs=se1):
                failed.append(a_["name"])
            y2 = F.conv2d(t_dev, w2q, padding=1)
            if psc is not None:
                y2 = y2 + F.conv2d(x, _round_t(torch.from_numpy(wsc * ss[:, None]), dt)[:, :, None, None])
            y2 = y2 + torch.from_numpy(shift2.astype(np.float32))[None, :, None, None]
            if psc is None:
                y2 = y2 + x
            se2 = 2.0 ** -16 * float(np.abs(shift2).max()) if dt == BF16 else 0.0
            if not _check(b_["name"], got, torch.relu(y2), dt, report, extra_abs=se2):
                failed.append(b_["name"])
            seen_forms.add(("block", xr.W, psc is not None))
            host.retire(i)
            continue
        if k not in ("conv", "deconv", "meta"):
            exe.forward(fr, only=i, dev=dev)
            if isinstance(st.get("out"), TRef):
                host.invalidate(st["out"])
            continue
        x = host.get(st["x"])
        if st.get("x2") is not None:      # conv over the virtual concat [x | x2 | zero padding of x2's buffer] (rd_conv3x3_bn_act_cat)
            x2 = host.get(st["x2"])
            x = torch.cat([x, x2, torch.zeros(x2.shape[0], st["x2"].cs - x2.shape[1], x2.shape[2], x2.shape[3])], 1)
        res = host.get(st["res"]) if st.get("res") is not None else None
        sx = host.get(st["sc_x"]) if st.get("sc_x") is not None else None
        exe.forward(fr, only=i, dev=dev)
        if isinstance(st.get("out"), TRef):
            host.invalidate(st["out"])
        name = st.get("name", "meta unit")
        if k == "meta":
            coord = torch.from_numpy(np.ascontiguousarray(fr[st["coord"]], dtype=np.float32))
            got = host.get(st["out"], fresh=True)
            u = 2.0 ** -9 if dt == BF16 else 2.0 ** -12
            zmax, zsq, n, nover = 0.0, 0.0, 0, 0
            for b in range(B):
                ref, sig = _meta_ref_and_sigma(x[b:b + 1], coord[b:b + 1], P, "res1_unit2", u)
                # the unit test's model (test_meta_kernel_unit), evaluated PER ELEMENT: sigma = the quadrature sum of the independent
                # 16-bit roundings inside the 576-term contraction, + one rounding of the output.  The rounding of a hidden unit enters
                # all 64 dynamic weights of its tap COHERENTLY, which the quadrature sum does not model: the tail is heavier than a
                # Gaussian's (9 sigma seen on 5e4 elements of the emulator build), so: rms within 1.5x the model, at most 1e-4 of the
                # elements beyond 7 sigma, none beyond 28 sigma (a mis-addressed halo pixel is an error of >= 100 sigma)
                tol = 7.0 * sig + u * ref.abs() + 1e-5
                err = (got[b:b + 1] - ref).abs()
                z = err / tol
                zmax = max(zmax, float(z.max()))
                nover += int((z > 1.0).sum())
                zsq += float(((got[b:b + 1] - ref) ** 2 / (sig ** 2 + (u * ref.abs()) ** 2 / 3 + 1e-12)).sum())
                n += ref.numel()
            zr = np.sqrt(zsq / n)
            report.append((name + " (rms err / model sigma %.2f; > 7 sigma: %d)" % (zr, nover), tuple(got.shape), zmax / 4.0, int(zmax > 4.0)))
            if not (zr < 1.5 and nover <= 1e-4 * n and zmax <= 4.0):
                failed.append(name + " (rms err / model sigma %.3f, worst |err| / (7 sigma) %.3f, %d of %d beyond 7 sigma)" % (zr, zmax, nover, n))
            host.retire(i)
            continue
        cout = st["cout"]
        s, t = bn_affine(P, st["bn"], st["eps"])
        if k == "deconv":
            assert st.get("fold"), "production deconvs carry the folded scale"
            w = np.asarray(P[name + "_weight"], np.float32)                       # (cin, cout, kh, kw)
            wq = _round_t(torch.from_numpy(w * s[None, :, None, None]), dt)
            y = F.conv_transpose2d(x, wq, stride=(1, st["stride_w"]), padding=(1, st["pad_w"]))
            y = y + torch.from_numpy(t)[None, :, None, None]
            assert st["flags"] == (R.RD_RELU_PRE | R.RD_ADD)
            ref = torch.relu(y) + res
            got = host.get(st["out"], fresh=True)
            form = ("deconv", st["cin"], cout, x.shape[3], st["stride_w"])
        else:
            assert st.get("ex") and st.get("fold"), "production 16-bit convs are 3x3 with the folded scale"
            w = _expand(np.asarray(P[name + "_weight"], np.float32), st.get("cmap"))
            wq = _round_t(torch.from_numpy(w * s[:, None, None, None]), dt)
            y = F.conv2d(x, wq, stride=(1, st["stride_w"]), padding=1)
            shift = t.astype(np.float64)
            if st.get("sc"):
                sc = st["sc"]
                wsc = np.asarray(P[sc["name"] + "_weight"], np.float32).reshape(cout, -1)
                wsc = _expand(wsc, sc.get("cmap"))
                ss, ts = bn_affine(P, sc["bn"], sc["eps"])
                wscq = _round_t(torch.from_numpy(wsc * ss[:, None]), dt)
                y = y + F.conv2d(sx, wscq[:, :, None, None], stride=(1, st["stride_w"]))
                shift = shift + ts
            y = y + torch.from_numpy(shift.astype(np.float32))[None, :, None, None]
            fl = st["flags"]
            if fl & R.RD_RELU_PRE:
                y = torch.relu(y)
            if res is not None:
                y = y + res
            if fl & R.RD_RELU_POST:
                y = torch.relu(y)
            form = ("conv", st["cin"], cout, x.shape[3], st["stride_w"], bool(st.get("sc")), res is not None, bool(st.get("head")),
                    st["out"].cs if isinstance(st.get("out"), TRef) else 0)
            if st.get("head"):
                # the tower's last conv + its 1x1 output conv in one launch: the 128-channel activation is rounded to the type,
                # then out = W . act + bias with fp32-accurate weights (hi + lo pair); head/builder.py:242-255
                h = st["head"]
                r0, r1 = h["rows"]
                hw = torch.from_numpy(np.asarray(P[h["name"] + "_weight"], np.float32).reshape(-1, cout)[r0:r1].copy())
                hb = torch.from_numpy(np.asarray(P[h["name"] + "_bias"], np.float32)[r0:r1].copy())
                act = _round_t(y, dt)
                ref = torch.einsum("oc,bchw->bhwo", hw, act).reshape(B, -1, r1 - r0) + hb
                flat = torch.from_numpy(exe.read_flat(h["out"]))
                flat = flat.reshape(B, h["N"], -1)
                got = flat[:, h["n_off"]:h["n_off"] + x.shape[2] * x.shape[3]]
                # an activation within fp32 noise of a rounding boundary may round the other way: one unit of the activation
                # times its weight -- a few such flips per output at most
                u = 2.0 ** -8 if dt == BF16 else 2.0 ** -11
                extra = 4 * u * float(act.abs().max()) * float(hw.abs().max())
                err = (got - ref).abs()
                tol = 1e-5 * max(1.0, float(ref.abs().max())) + extra
                worst = float(err.max()) / tol
                report.append((name + " + " + h["name"], tuple(ref.shape), worst, int((err > tol).sum())))
                if worst > 1.0:
                    failed.append(name)
                seen_forms.add(form)
                host.retire(i)
                continue
            ref = y
            got = host.get(st["out"], fresh=True)
        seen_forms.add(form)
        shift_err = 2.0 ** -16 * float(np.abs(t).max()) if dt == BF16 else 0.0
        if not _check(name, got, ref, dt, report, extra_abs=shift_err):
            failed.append(name)
        host.retire(i)
    dtn = "bf16" if dt == BF16 else "fp16"
    print("\nper-step tight parity at B = 8, %s, %s (worst |err| / tolerance, elements over):" % ("64 x 2048 x 5 two-class" if variant == "kitti" else "64 x 2656", dtn))
    for name, shape, worst, nbad in report:
        print("  %-44s %-22s %.3f  %d" % (name, "x".join(str(v) for v in shape), worst, nbad))
    print("%d distinct launch forms, %d steps checked, %.0f s" % (len(seen_forms), len(report), time.time() - t00))
    kinds = [s["kind"] for s in plan.steps]
    # (a fused BasicBlock reports three lines: bit-equality with its two launches, conv1, conv2)
    assert kinds.count("conv") + 2 * kinds.count("block") + kinds.count("deconv") + kinds.count("meta") == 78 and kinds.count("block") == 8
    assert len(report) == kinds.count("conv") + 3 * kinds.count("block") + kinds.count("deconv") + kinds.count("meta")
    assert not failed, failed
