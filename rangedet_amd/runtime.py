"""Executor: binds a lowered Plan (rangedet_amd.lower) to device buffers and parameters and replays it as C-ABI calls
into librangedet_hip.so on one HIP stream.  PyTorch is used only as the device allocator / stream provider.

This is the build's counterpart of the reference's DetModule.bind/set_params/forward/get_outputs
(utils/detection_module.py:419-510,627-679,783-805) for the test symbol: ``Executor(plan, params).forward(inputs)``
returns the Group outputs [rec_id, fg_cls_score, decoded_bbox, zeros, gt_bbox_imu, gt_class] (builder.py:77).
"""
import os

import numpy as np

from . import devswitch, lib as rdlib
from .lower import FlatRef, TRef


class TorchAllocator:
    """Device memory + stream through torch (ROCm).  Fails loudly without a GPU -- there is no CPU path."""

    def __init__(self, device="cuda:0", stream=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("rangedet_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.torch = torch
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self._stream = stream

    @property
    def stream(self):
        st = self._stream or self.torch.cuda.current_stream(self.device)
        return st.cuda_stream

    def alloc(self, nbytes, zero=False):
        f = self.torch.zeros if zero else self.torch.empty
        return f(max(int(nbytes), 16), dtype=self.torch.uint8, device=self.device)

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        t = self.torch.from_numpy(a.view(np.uint8).reshape(-1)).to(self.device)
        return t

    def ptr(self, buf):
        return buf.data_ptr()

    def view_f32(self, buf, shape):
        n = int(np.prod(shape))
        return buf[: n * 4].view(self.torch.float32).view(*shape)

    def view_i32(self, buf, shape):
        n = int(np.prod(shape))
        return buf[: n * 4].view(self.torch.int32).view(*shape)

    def to_numpy(self, t):
        return t.detach().cpu().numpy()

    def as_device_f32(self, x):
        """numpy / torch input -> contiguous float32 device tensor (no copy if it already is one)."""
        t = self.torch
        if isinstance(x, np.ndarray):
            return t.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(self.device)
        return x.to(device=self.device, dtype=t.float32).contiguous()

    def assign(self, dst_view, src):
        dst_view.copy_(src.reshape(dst_view.shape))

    # ---- streams / events (the post-processing of frame i overlaps the forward of frame i+1) ----------------
    def new_stream(self, priority=0):
        """priority: 0 = normal, -1 = high (torch's convention; the HIP runtime maps it onto the hardware queue's priority)."""
        return self.torch.cuda.Stream(device=self.device, priority=priority)

    def stream_ptr(self, stream):
        return stream.cuda_stream if stream is not None else self.stream

    def record_event(self, stream=None):
        ev = self.torch.cuda.Event()
        ev.record(stream if stream is not None else self.torch.cuda.current_stream(self.device))
        return ev

    def wait_event(self, ev, stream=None):
        (stream if stream is not None else self.torch.cuda.current_stream(self.device)).wait_event(ev)

    def sync(self):
        self.torch.cuda.synchronize(self.device)


def bn_affine(P, name, eps, pad_to=None):
    """inference BatchNorm as scale / shift; pad_to: zero-padded output channels of a layer that runs on a wider kernel (scale 1, shift 0)"""
    g, b, m, v = (np.asarray(P[name + s], np.float64) for s in ("_gamma", "_beta", "_moving_mean", "_moving_var"))
    s = g / np.sqrt(v + eps)
    s, t = s.astype(np.float32), (b - m * s).astype(np.float32)
    if pad_to is not None and pad_to > len(s):
        s = np.concatenate([s, np.ones(pad_to - len(s), np.float32)])
        t = np.concatenate([t, np.zeros(pad_to - len(t), np.float32)])
    return s, t


def pad_rows(w, n, axis=0):
    """weights with zero rows appended along `axis` up to n (a layer whose logical width runs zero-padded on a 64 / 128-channel kernel)"""
    if w.shape[axis] >= n:
        return w
    shp = list(w.shape)
    shp[axis] = n - w.shape[axis]
    return np.concatenate([w, np.zeros(shp, w.dtype)], axis)


class Executor:
    def __init__(self, plan, params, lib=None, alloc=None, reuse_buffers=True, keep_sorted_idx=False):
        """keep_sorted_idx: also keep get_sorted_foreground's (B,k) flat indices (parity tests read them with sorted_idx())."""
        self.keep_sorted_idx = keep_sorted_idx
        self.plan = plan
        self.lib = lib or rdlib.get_lib()
        self.alloc = alloc or TorchAllocator()
        self.dtype = plan.dtype
        self.B = plan.batch
        self._phys = {}
        self._assign_buffers(reuse_buffers)
        self._bound = [self._bind(st, params) for st in plan.steps]

    # ---- memory ------------------------------------------------------------------------------------------------
    def _assign_buffers(self, reuse):
        plan = self.plan
        last_use, first_def = {}, {}
        for i, st in enumerate(plan.steps):
            for v in st.values():
                if isinstance(v, (TRef, FlatRef)):
                    last_use[v.buf] = i
                    first_def.setdefault(v.buf, i)
        for kind, v in plan.outputs:
            if kind in ("flat", "flat_i32"):
                last_use[v.buf] = len(plan.steps) + 1
        free = {}
        release = {}
        for b, i in last_use.items():
            release.setdefault(i, []).append(b)
        order = sorted(plan.buffers, key=lambda b: first_def.get(b, 0))
        by_def = {}
        for b in order:
            by_def.setdefault(first_def.get(b, 0), []).append(b)
        for i in range(len(plan.steps) + 2):
            for b in by_def.get(i, []):
                info = plan.buffers[b]
                if reuse and not info["persistent"] and free.get(info["nbytes"]):
                    self._phys[b] = free[info["nbytes"]].pop()
                else:
                    self._phys[b] = self.alloc.alloc(info["nbytes"], zero=info["zero"])
            for b in release.get(i, []):
                info = plan.buffers[b]
                if reuse and not info["persistent"] and last_use[b] <= len(plan.steps):
                    free.setdefault(info["nbytes"], []).append(self._phys[b])
        self.device_bytes = sum(int(t.numel()) if hasattr(t, "numel") else len(t) for t in {id(v): v for v in self._phys.values()}.values())

    def p(self, ref):
        return self.alloc.ptr(self._phys[ref.buf])

    # ---- parameter binding -------------------------------------------------------------------------------------------
    def _bind(self, st, P):
        L, A, dt = self.lib, self.alloc, self.dtype
        k = st["kind"]
        b = dict(st)
        if k == "conv_pair":
            b["a"], b["b"] = self._bind(st["a"], P), self._bind(st["b"], P)
            return b
        if k == "block":   # a fused BasicBlock (lower._fuse_blocks): both 3x3 weights in one image, the shortcut's as for the unfused conv2
            ca, cb = st["a"], st["b"]
            w1, w2 = (np.asarray(P[c["name"] + "_weight"], np.float32) for c in (ca, cb))
            if ca.get("cmap"):   # (the first block's input buffer carries alignment padding: zero weight columns there)
                wp = np.zeros((w1.shape[0], len(ca["cmap"])) + w1.shape[2:], np.float32)
                for pc, lc in enumerate(ca["cmap"]):
                    if lc >= 0:
                        wp[:, pc] = w1[:, lc]
                w1 = wp
            (s1, t1), (s2, t2) = bn_affine(P, ca["bn"], ca["eps"]), bn_affine(P, cb["bn"], cb["eps"])
            m16 = bool(st.get("m16")) and w1.shape[1] == 64      # the 16 x 16 x 32 MFMA form (lower._mark_mfma16): its own weight images
            b["m16"] = m16
            b["w"] = A.upload(L.pack_block64(w1, s1, w2, s2, dtype=dt, m16=m16))
            b["shift1"] = A.upload(t1)
            b["sc_w"] = None
            if cb.get("sc"):
                sc = cb["sc"]
                wsc = np.asarray(P[sc["name"] + "_weight"], np.float32).reshape(64, -1)
                if sc.get("cmap"):
                    wq = np.zeros((64, len(sc["cmap"])), np.float32)
                    for pc, lc in enumerate(sc["cmap"]):
                        if lc >= 0:
                            wq[:, pc] = wsc[:, lc]
                    wsc = wq
                ss, ts = bn_affine(P, sc["bn"], sc["eps"])
                b["sc_w"] = A.upload(L.pack_conv1x1_sc_m16(wsc, ss, dtype=dt) if m16 else L.pack_conv1x1_sc(wsc, fold_scale=ss, dtype=dt))
                t2 = (t2.astype(np.float64) + ts).astype(np.float32)
            b["shift2"] = A.upload(t2)
            b["cin"] = w1.shape[1]
            return b
        if k == "conv":
            w = pad_rows(np.asarray(P[st["name"] + "_weight"], np.float32), st["cout"])
            if st.get("cmap"):   # the input is a concat buffer with alignment padding: zero weight columns there
                wp = np.zeros((w.shape[0], len(st["cmap"])) + w.shape[2:], np.float32)
                for pc, lc in enumerate(st["cmap"]):
                    if lc >= 0:
                        wp[:, pc] = w[:, lc]
                w = wp
            s, t = bn_affine(P, st["bn"], st["eps"], pad_to=st["cout"])
            if st.get("sc"):      # fused projection shortcut: both BN scales folded into the weights, one shift for the sum
                sc = st["sc"]
                wsc = np.asarray(P[sc["name"] + "_weight"], np.float32)
                wsc = pad_rows(wsc.reshape(wsc.shape[0], -1), st["cout"])
                if sc.get("cmap"):
                    wq = np.zeros((w.shape[0], len(sc["cmap"])), np.float32)
                    for pc, lc in enumerate(sc["cmap"]):
                        if lc >= 0:
                            wq[:, pc] = wsc[:, lc]
                    wsc = wq
                ss, ts = bn_affine(P, sc["bn"], sc["eps"], pad_to=st["cout"])
                b["w"] = A.upload(L.pack_conv3x3_ex(w, st["stride_w"], st["x"].cs, fold_scale=s, dtype=dt))
                b["sc_w"] = A.upload(L.pack_conv1x1_sc(wsc, fold_scale=ss, dtype=dt))
                b["scale"], b["shift"] = None, A.upload((t.astype(np.float64) + ts).astype(np.float32))
            elif st.get("x2") is not None:          # conv over the virtual concat [x | x2] (lower._concat)
                b["w"] = A.upload(L.pack_conv3x3_cat(w, st["x"].C, st["x2"].cs, fold_scale=s, dtype=dt))
                b["scale"], b["shift"] = None, A.upload(t)
                b["flags"] = st["flags"] | rdlib.RD_SCALE_FOLDED
            elif st.get("ex") and st.get("fold") and st.get("m16") and \
                    L.raw("rd_conv3x3_mfma16_ok")(w.shape[1], st["cout"], 1, st["x"].W, 1 if st.get("head") else 0) == 1:
                # the same in the 16 x 16 x 32 MFMA form (lower._mark_mfma16)
                b["w"] = A.upload(L.pack_conv3x3_m16(w, s, dtype=dt))
                b["scale"], b["shift"] = None, A.upload(t)
                b["flags"] = st["flags"] | rdlib.RD_SCALE_FOLDED | rdlib.RD_MFMA16
            elif st.get("ex") and st.get("fold"):   # scale folded into the weights, the shift enters through the accumulators
                b["w"] = A.upload(L.pack_conv3x3_ex(w, st["stride_w"], st["x"].cs, fold_scale=s, dtype=dt))
                b["scale"], b["shift"] = None, A.upload(t)
                b["flags"] = st["flags"] | rdlib.RD_SCALE_FOLDED
            elif st.get("ex"):
                b["w"] = A.upload(L.pack_conv3x3_ex(w, st["stride_w"], st["x"].cs, dtype=dt))
                b["scale"], b["shift"] = A.upload(s), A.upload(t)
            else:
                b["w"] = A.upload(L.pack_conv_weight(w, dt))
                b["scale"], b["shift"] = A.upload(s), A.upload(t)
        elif k == "deconv":
            w = pad_rows(np.asarray(P[st["name"] + "_weight"], np.float32), st["cout"], axis=1)      # (cin, cout, kh, kw)
            s, t = bn_affine(P, st["bn"], st["eps"], pad_to=st["cout"])
            fs = s if st.get("fold") else None
            imgs = [L.pack_deconv_weight(w, st["stride_w"], st["pad_w"], ph, dt, fold_scale=fs) for ph in range(st["stride_w"])]
            # all phases in ONE launch when the library has that form for this layer (16-bit, folded scale, 3 x 2 phases):
            # the phase images back to back in one buffer (RD_DECONV_PER_PHASE=1: one launch per phase, for A/B runs)
            b["all_phases"] = bool(st.get("one_launch")) and len({len(i) for i in imgs}) == 1 and \
                L.raw("rd_deconv2d_all_phases_ok")(st["k"][0], st["k"][1], st["stride_w"], st["pad_w"], st["cout"], dt) == 1
            if st.get("one_launch") and not b["all_phases"] and not L.raw("rd_deconv2d_all_phases_ok")(st["k"][0], st["k"][1], st["stride_w"], st["pad_w"], st["cout"], dt):
                b["one_launch"] = False      # (a development switch turned the library's form off: per-phase launches)
            if b["all_phases"]:
                b["w_all"], b["w_pb"] = A.upload(np.concatenate(imgs)), len(imgs[0])
            # phases 2p | 2p+1 as one 128-channel problem where the layer has that form and its tensors are dense (agg1: 128 -> 64, stride 4;
            # RD_DECONV_NO_PAIRS=1: the all-phases launch, for A/B runs)
            r_, o_ = st["res"], st["out"]
            b["pairs"] = b["all_phases"] and not devswitch.get("RD_DECONV_NO_PAIRS") and o_.cs == st["cout"] and o_.co == 0 and \
                (r_ is None or (r_.cs == st["cout"] and r_.co == 0)) and \
                L.raw("rd_deconv2d_phase_pairs_ok")(st["k"][0], st["k"][1], st["stride_w"], st["pad_w"], st["cout"], dt) == 1
            if b["pairs"]:
                pimgs = [L.pack_deconv_phase_pair(imgs[p], imgs[p + 1], st["cin"], dt) for p in range(0, st["stride_w"], 2)]
                b["w_pairs"], b["w_pairb"], b["shift2"] = A.upload(np.concatenate(pimgs)), len(pimgs[0]), A.upload(np.concatenate([t, t]))
            b["w"] = [A.upload(i) for i in imgs]
            b["scale"], b["shift"] = (None if st.get("fold") else A.upload(s)), A.upload(t)
            if st.get("fold"):
                b["flags"] = st["flags"] | rdlib.RD_SCALE_FOLDED
        elif k == "meta":
            s1, t1 = bn_affine(P, st["bn1"], st["eps1"])
            s2, t2 = bn_affine(P, st["bn2"], st["eps2"])
            pk = L.pack_meta(P[st["mlp0"] + "_weight"].reshape(32, 3), P[st["mlp0"] + "_bias"],
                             P[st["mlp1"] + "_weight"].reshape(64, 32), P[st["mlp1"] + "_bias"], s1, t1,
                             P[st["agg"] + "_weight"].reshape(64, 576), s2, t2, dt)
            b["packed"] = A.upload(pk)
        if k == "conv" and st.get("head"):
            h = st["head"]
            r0, r1 = h["rows"]
            hw = np.asarray(P[h["name"] + "_weight"], np.float32).reshape(-1, st["cout"])[r0:r1]
            b["head_w"] = A.upload(L.pack_head_weight(hw, dtype=dt, m16=bool(b.get("flags", 0) & rdlib.RD_MFMA16)))
            b["head_bias"] = A.upload(np.asarray(P[h["name"] + "_bias"], np.float32)[r0:r1])
        elif k == "head_out":
            r0, r1 = st["rows"]
            w = np.asarray(P[st["name"] + "_weight"], np.float32).reshape(-1, st["x"].C)[r0:r1]
            b["w"] = A.upload(w)
            b["bias"] = A.upload(np.asarray(P[st["name"] + "_bias"], np.float32)[r0:r1])
        elif k == "nms3d":
            nb = L.raw("rd_nms3d_workspace_bytes")(st["N"], self.B)
            b["ws"] = A.alloc(nb)
            b["ws_bytes"] = nb
        elif k == "sorted_fg":
            nb = L.raw("rd_sorted_foreground_workspace_bytes")(st["N"], st["k"]) * self.B
            b["ws"] = A.alloc(nb)
            b["ws_bytes"] = nb
            b["idx"] = A.alloc(self.B * st["k"] * 4) if self.keep_sorted_idx else None
        return b

    # ---- execution ------------------------------------------------------------------------------------------------------
    def forward(self, inputs, only=None, dev=None):
        """inputs: name -> float32 array/tensor WITH batch dim (the reference's data_name list, config:400-404).
        `only` (profiling aid): run just that plan step index."""
        L, A, dt, B = self.lib, self.alloc, self.dtype, self.B
        st_ = A.stream
        dev = {} if dev is None else dev

        def din(name):
            if name not in dev:
                dev[name] = A.as_device_f32(inputs[name])
            return dev[name]

        steps = self._bound if only is None else [self._bound[only]]
        for b in steps:
            k = b["kind"]
            if k == "nchw_in":
                o = b["out"]
                src = din(b["name"])
                assert tuple(src.shape) == (B, o.C, o.H, o.W), (b["name"], tuple(src.shape), (B, o.C, o.H, o.W))
                L.call("rd_nchw_to_nhwc", A.ptr(src), self.p(o), B, o.C, o.H, o.W, o.cs, o.co, b["zero_pad"], dt, st_)
            elif k == "block":
                x, o = b["x"], b["out"]
                if b.get("m16"):
                    L.call("rd_block64_m16_bn_act", self.p(x), x.cs, x.co, A.ptr(b["w"]), A.ptr(b["shift1"]), A.ptr(b["shift2"]),
                           A.ptr(b["sc_w"]) if b["sc_w"] is not None else None, self.p(o), o.cs, o.co, B, x.H, x.W, dt, st_)
                else:
                    L.call("rd_block64_bn_act", self.p(x), x.cs, x.co, b["cin"], A.ptr(b["w"]), A.ptr(b["shift1"]), A.ptr(b["shift2"]),
                           A.ptr(b["sc_w"]) if b["sc_w"] is not None else None, self.p(o), o.cs, o.co, B, x.H, x.W, dt, st_)
            elif k == "conv_pair":
                # two convs of one shape in ONE launch (lower._pair_equal_convs: the cls and reg tower conv of a head level)
                p, q = b["a"], b["b"]
                x0, x1 = p["x"], q["x"]
                cin = len(p["cmap"]) if p.get("cmap") else p["cin"]
                if p.get("head"):
                    h0, h1 = p["head"], q["head"]
                    L.call("rd_conv2d_bn_act_head_out_pair",
                           self.p(x0), x0.co, A.ptr(p["w"]), A.ptr(p["shift"]), A.ptr(p["head_w"]), A.ptr(p["head_bias"]),
                           self.p(h0["out"]), h0["N"] * h0["nout"], h0["nout"],
                           self.p(x1), x1.co, A.ptr(q["w"]), A.ptr(q["shift"]), A.ptr(q["head_w"]), A.ptr(q["head_bias"]),
                           self.p(h1["out"]), h1["N"] * h1["nout"], h1["nout"],
                           x0.cs, h0["n_off"], B, x0.H, x0.W, cin, p["flags"], dt, st_)
                else:
                    o0, o1 = p["out"], q["out"]
                    L.call("rd_conv3x3_bn_act_pair", self.p(x0), x0.co, A.ptr(p["w"]), A.ptr(p["shift"]), self.p(o0), o0.co,
                           self.p(x1), x1.co, A.ptr(q["w"]), A.ptr(q["shift"]), self.p(o1), o1.co, x0.cs, o0.cs, B, x0.H, x0.W, cin,
                           p["flags"], dt, st_)
            elif k == "conv" and b.get("head"):
                x, h = b["x"], b["head"]
                L.call("rd_conv2d_bn_act_head_out", self.p(x), x.cs, x.co, A.ptr(b["w"]),
                       A.ptr(b["scale"]) if b["scale"] is not None else None, A.ptr(b["shift"]), B,
                       x.H, x.W, b["cin"], b["flags"], A.ptr(b["head_w"]), A.ptr(b["head_bias"]), self.p(h["out"]),
                       h["N"] * h["nout"], h["n_off"], h["nout"], dt, st_)
            elif k == "conv" and b.get("x2") is not None:
                # conv over the virtual concat [x | x2] (lower._concat): neither a shared buffer nor a copy exists
                x, x2, o = b["x"], b["x2"], b["out"]
                L.call("rd_conv3x3_bn_act_cat", self.p(x), x.cs, x.co, x.C, self.p(x2), x2.cs, x2.co, x2.cs, A.ptr(b["w"]), A.ptr(b["shift"]),
                       self.p(o), o.cs, o.co, B, x.H, x.W, b["cout"], b["flags"], dt, st_)
            elif k == "conv" and b.get("ex"):
                x, o, r, sx = b["x"], b["out"], b["res"], b.get("sc_x")
                cin = len(b["cmap"]) if b.get("cmap") else b["cin"]
                L.call("rd_conv3x3_bn_act_ex", self.p(x), x.cs, x.co, A.ptr(b["w"]), A.ptr(b["scale"]) if b["scale"] is not None else None,
                       A.ptr(b["shift"]), self.p(r) if r else None, r.cs if r else 0, r.co if r else 0,
                       self.p(sx) if sx else None, sx.cs if sx else 0, sx.co if sx else 0,
                       (len(b["sc"]["cmap"]) if b["sc"].get("cmap") else b["sc"]["cin"]) if sx else 0,
                       A.ptr(b["sc_w"]) if sx else None, self.p(o), o.cs, o.co, B, x.H, x.W, cin, b["cout"], b["stride_w"],
                       b["flags"], dt, st_)
            elif k == "conv":
                x, o, r = b["x"], b["out"], b["res"]
                L.call("rd_conv2d_bn_act", self.p(x), x.cs, x.co, A.ptr(b["w"]), A.ptr(b["scale"]), A.ptr(b["shift"]),
                       self.p(r) if r else None, r.cs if r else 0, r.co if r else 0, self.p(o), o.cs, o.co, B, x.H, x.W,
                       b["cin"], b["cout"], b["k"][0], b["k"][1], b["stride_w"], b["flags"], dt, st_)
            elif k == "deconv":
                x, o, r = b["x"], b["out"], b["res"]
                if b.get("pairs"):
                    L.call("rd_deconv2d_bn_act_pairs", self.p(x), x.cs, x.co, A.ptr(b["w_pairs"]), b["w_pairb"], A.ptr(b["shift2"]), self.p(r), r.cs,
                           r.co, self.p(o), o.cs, o.co, B, x.H, x.W, b["cin"], b["cout"], b["k"][0], b["k"][1], b["stride_w"], b["pad_w"],
                           b["flags"], dt, st_)
                    continue
                if b.get("all_phases"):
                    L.call("rd_deconv2d_bn_act_all", self.p(x), x.cs, x.co, A.ptr(b["w_all"]), b["w_pb"], A.ptr(b["shift"]), self.p(r), r.cs,
                           r.co, self.p(o), o.cs, o.co, B, x.H, x.W, b["cin"], b["cout"], b["k"][0], b["k"][1], b["stride_w"], b["pad_w"],
                           b["flags"], dt, st_)
                    continue
                for ph in range(b["stride_w"]):
                    L.call("rd_deconv2d_bn_act", self.p(x), x.cs, x.co, A.ptr(b["w"][ph]),
                           A.ptr(b["scale"]) if b["scale"] is not None else None, A.ptr(b["shift"]), self.p(r), r.cs, r.co, self.p(o), o.cs, o.co, B, x.H, x.W, b["cin"],
                           b["cout"], b["k"][0], b["k"][1], b["stride_w"], b["pad_w"], ph, b["flags"], dt, st_)
            elif k == "meta":
                x, o = b["x"], b["out"]
                c = din(b["coord"])
                assert tuple(c.shape) == (B, 3, x.H, x.W)
                L.call("rd_meta_kernel_fwd", self.p(x), x.cs, x.co, A.ptr(c), A.ptr(b["packed"]), self.p(o), o.cs, o.co,
                       B, x.H, x.W, dt, st_)
            elif k == "head_out":
                x, o = b["x"], b["out"]
                L.call("rd_head_out", self.p(x), x.cs, x.co, A.ptr(b["w"]), A.ptr(b["bias"]), self.p(o),
                       b["N"] * b["nout"], b["n_off"], B, x.H, x.W, x.C, b["nout"], dt, st_)
            elif k == "concat_in":
                o = b["out"]
                row = b["N"] * b["last"] * 4                      # bytes of one frame's concatenated row
                off = 0
                for name, n in b["names"]:
                    src = din(name)
                    assert tuple(src.shape)[:2] == (B, n), (name, tuple(src.shape))
                    L.call("rd_copy_rows", A.ptr(src), n * b["last"] * 4, self.p(o), row, off * b["last"] * 4,
                           n * b["last"] * 4, B, st_)
                    off += n
            elif k == "sorted_fg":
                L.call("rd_sorted_foreground", self.p(b["score"]), self.p(b["delta"]), self.p(b["pc"]), self.p(b["mask"]),
                       B, b["N"], b["k"], b["D"], b["apply_sigmoid"], self.p(b["out_score"]), self.p(b["out_delta"]),
                       self.p(b["out_pc"]), A.ptr(b["idx"]) if b["idx"] is not None else None, A.ptr(b["ws"]), b["ws_bytes"], st_)
            elif k == "decode":
                L.call("rd_decode3d_bbox", self.p(b["delta"]), self.p(b["pc"]), self.p(b["out"]), B, b["k"],
                       b["box_type"], b["is_bin"], st_)
            elif k == "batch_riou":
                g = din(b["gt"])
                three_d = b.get("iou_type", "bev") == "3d"
                assert tuple(g.shape) == (B, b["n_gt"], 7 if three_d else 8), (b["gt"], tuple(g.shape))
                L.call("rd_batch_rotated_iou_3d" if three_d else "rd_batch_rotated_iou", self.p(b["boxes"]), 10, A.ptr(g), self.p(b["out"]),
                       None, B, b["N"], b["n_gt"], st_)
            elif k == "nms3d":
                L.call("rd_nms3d", self.p(b["boxes"]), B, b["N"], b["thr"], b["max_keep"], b["normal_iou"], self.p(b["keep"]),
                       self.p(b["out"]), A.ptr(b["ws"]), b["ws_bytes"], st_)
            else:
                raise RuntimeError("unknown plan step %r" % k)
        if only is not None:
            return None
        outs = []
        for kind, v in self.plan.outputs:
            if kind == "flat":
                outs.append(A.view_f32(self._phys[v.buf], (B,) + tuple(v.shape)))
            elif kind == "flat_i32":
                outs.append(A.view_i32(self._phys[v.buf], (B,) + tuple(v.shape)))
            elif kind == "input":
                outs.append(inputs.get(v))
            else:
                outs.append(np.zeros(v, np.float32))
        return outs

    def read_flat(self, ref):
        """numpy copy of a flat float32 plan tensor (B, *shape), e.g. the pre-sort logits/deltas (tests only)."""
        self.alloc.sync()
        return np.array(self.alloc.to_numpy(self.alloc.view_f32(self._phys[ref.buf], (self.B,) + tuple(ref.shape))))

    def sorted_idx(self, which=0):
        """(B,k) int32 flat indices chosen by the which-th get_sorted_foreground step (needs keep_sorted_idx=True)."""
        b = [x for x in self._bound if x["kind"] == "sorted_fg"][which]
        self.alloc.sync()
        return np.array(self.alloc.to_numpy(self.alloc.view_i32(b["idx"], (self.B, b["k"]))))

    def debug_tensor(self, ref):
        """NCHW float32 numpy copy of an activation (tests only)."""
        A, L = self.alloc, self.lib
        out = A.alloc(self.B * ref.C * ref.H * ref.W * 4)
        L.call("rd_nhwc_to_nchw", self.p(ref), A.ptr(out), self.B, ref.C, ref.H, ref.W, ref.cs, ref.co, self.dtype, A.stream)
        A.sync()
        return A.to_numpy(A.view_f32(out, (self.B, ref.C, ref.H, ref.W)))
