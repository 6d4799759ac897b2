"""Lowering: recorded symbol graph (rangedet_amd.mx) -> plan of fused HIP launches (include/rangedet_hip.h).

Fusions (each replaces an MXNet op chain of the reference by one kernel):
  relu(BN(Convolution(x)))                         -> rd_conv2d_bn_act                         (dla_backbone.py:24-32, builder.py:221-240)
  relu(BN(Convolution(x)) + shortcut)              -> rd_conv2d_bn_act + RD_ADD|RD_RELU_POST   (dla_backbone.py:34-56)
  BN(Convolution 1x1)   (projection shortcut)      -> rd_conv2d_bn_act, no activation          (dla_backbone.py:44-51)
  bf16: relu(BN(Conv3x3(h)) + BN(Conv1x1(x)))      -> ONE rd_conv3x3_bn_act_ex: the projection shortcut is accumulated in the 3x3
                                                      kernel's epilogue; stride (1,2) 3x3 convs run on the pixel-pair view
  skip + relu(BN(Deconvolution(u)))                -> rd_deconv2d_bn_act per phase, RD_RELU_PRE|RD_ADD   (dla_backbone.py:117-126)
  meta_baseline_bias(...) + BN + relu + 1x1 + BN + relu   -> rd_meta_kernel_fwd                (meta_kernel.py:166-240, dla_backbone.py:92-97)
  concat(data, agg3)                               -> 16-bit: never written, the consuming convs read both tensors (rd_conv3x3_bn_act_cat);
                                                      fp32: producers write straight into one buffer  (dla_backbone.py:153-154)
  1x1 logit/delta conv + cast + reshape/slice/squeeze/transpose + concat over levels
                                                   -> rd_head_out into flat (B,N[,8]) buffers   (builder.py:242-261,99-154)
  sigmoid + Custom(get_sorted_foreground)          -> rd_sorted_foreground(apply_sigmoid=1)     (builder.py:459-461,512-521)
  contrib.Decode3DBbox                             -> rd_decode3d_bbox                          (builder.py:522-525)
  contrib.NMS3D (wnms=False)                       -> rd_nms3d                                  (builder.py:530-534)
  Custom(batch_rotated_iou, iou_type='bev' | '3d') -> rd_batch_rotated_iou / rd_batch_rotated_iou_3d  (builder.py:176-182)
Anything that does not match raises NotImplementedError at lowering time -- there is no slow generic path.
"""
import os
from dataclasses import dataclass, field

from . import devswitch
from .lib import H16, RD_ADD, RD_BF16, RD_F16, RD_F32, RD_RELU_POST, RD_RELU_PRE


@dataclass
class TRef:
    """Channels-last activation [B][H][W][cs], channels [co, co+C) of logical buffer `buf`."""
    buf: int
    C: int
    H: int
    W: int
    cs: int
    co: int = 0
    cmap: tuple = None   # physical channel -> logical channel (-1 = zero padding) when a concat had to align its parts
    tail: "TRef" = None  # 16-bit "virtual concat": the channels after this tensor's C come from THAT tensor (its cs channels);
    #                      the concatenation is never written, the consuming conv reads both (rd_conv3x3_bn_act_cat)


@dataclass
class FlatRef:
    """Flat float32 device tensor (B, *shape)."""
    buf: int
    shape: tuple


@dataclass
class Plan:
    dtype: int
    batch: int
    steps: list = field(default_factory=list)
    buffers: dict = field(default_factory=dict)   # logical id -> dict(nbytes, persistent, zero)
    outputs: list = field(default_factory=list)   # per Group output: ("flat", FlatRef) | ("input", name) | ("zeros", shape)
    input_vars: dict = field(default_factory=dict)  # var name -> shape (without batch) expected from the caller
    num_classes: int = 1


def conv_steps(steps):
    """The conv / deconv steps of a plan one by one: the two members of a "conv_pair" (one launch) as two entries.
    Likewise the two convs of a fused BasicBlock ("block").
    Yields (step, launches): launches = what the step adds to the launch count (1, 0.5 + 0.5 for a pair; a transposed conv: 1 when
    all its phases run in one launch, else stride_w)."""
    for st in steps:
        if st["kind"] in ("conv_pair", "block"):   # (a fused BasicBlock: its two convs, one launch)
            yield st["a"], 0.5
            yield st["b"], 0.5
        elif st["kind"] == "conv":
            yield st, 1
        elif st["kind"] == "deconv":
            yield st, 1 if st.get("one_launch") else st["stride_w"]


def _strip_cast(s):
    while s.op == "cast":
        s = s.inputs[0]
    return s


def _is(s, op, **attrs):
    return s.op == op and all(s.attrs.get(k) == v for k, v in attrs.items())


class Lowering:
    def __init__(self, group, input_shapes, dtype, batch):
        """input_shapes: var name -> shape WITHOUT the batch dim, e.g. input_data: (8,64,2656)."""
        assert dtype in (RD_F32, RD_BF16, RD_F16)
        self.dtype = dtype
        self.h16 = dtype in H16                    # 16-bit activations / weights: the persistent-kernel fusions apply
        self.esz = 2 if self.h16 else 4
        self.gran = 16 if self.h16 else 8          # channel granule = one MFMA k-step
        self.B = batch
        self.shapes = dict(input_shapes)
        self.plan = Plan(dtype=dtype, batch=batch)
        self.memo = {}
        self.nbuf = 0
        self.consumers = {}
        self._count_consumers(group)
        for out in group.inputs:
            self.plan.outputs.append(self.emit_value(out))
        self._fuse_head_out()
        self._fuse_blocks()
        self._pair_equal_convs()
        self._mark_mfma16()

    # ---- fusion ------------------------------------------------------------------------------------------------
    def _mark_mfma16(self):
        """16-bit: the stride-1 3x3 convs with 128 output channels and no fused shortcut (tower conv_1 / conv_2 of every head level, conv_0
        of the levels whose input is 64 / 128 channels, the last tower convs with their fused output conv, the 128-channel BasicBlock
        convs of the backbone -- head/builder.py:221-240, dla_backbone.py:18-56) are launched in the v_mfma_f32_16x16x32 form of the
        persistent kernel (RD_MFMA16, k_conv3.h M16): the conv's numbers bit for bit (a fused output conv sums in another order: fp32
        rounding level), 3 - 6 % less time from W = 664 up because the part sustains a higher clock on that instruction under its power
        cap; likewise the fused 64-channel BasicBlocks with 64 input channels (rd_block64_m16_bn_act,
        bit-identical).  Step key "m16" is a request: the executor asks rd_conv3x3_mfma16_ok, packs with rd_pack_conv3x3_m16_host /
        rd_pack_head_weight_m16_host / rd_pack_block64_m16_host and passes the flag / calls that entry.
        RD_NO_MFMA16=1 (development switch): off; RD_NO_MFMA16_BLOCK=1: only the blocks off."""
        if not self.h16 or devswitch.get("RD_NO_MFMA16"):
            return
        for st in self.plan.steps:
            if st["kind"] == "block" and st["a"]["cin"] == 64 and not devswitch.get("RD_NO_MFMA16_BLOCK"):
                st["m16"] = True          # the fused 64-channel BasicBlocks too (rd_block64_m16_bn_act; not the network's first block)
            if st["kind"] == "conv" and tuple(st["k"]) == (3, 3) and st["cout"] == 128 and st["stride_w"] == 1 and st.get("ex") and \
                    st.get("fold") and not st.get("sc") and st.get("x2") is None and not st.get("s2view"):
                cin = len(st["cmap"]) if st.get("cmap") else st["cin"]
                if cin % 32 == 0 and cin >= 32:
                    st["m16"] = True

    def _pair_equal_convs(self):
        """16-bit: two 3x3 convs of the SAME shape whose inputs are both ready run as ONE launch (rd_conv3x3_bn_act_pair /
        rd_conv2d_bn_act_head_out_pair) -- in this graph the cls and the reg tower conv i of every head level
        (builder.py:221-240: the towers are built one after the other, layer i of both only depends on layer i-1 of its own
        tower).  The later conv moves up to the earlier one's place in the plan; it may only do so when the tensor it reads was
        written before that place.  Plan step kind "conv_pair": a / b = the two conv steps as they were."""
        # OFF by default (RD_PAIR=1 turns it on, RD_PAIR_MAXW limits it to narrow levels).  Measured on one box, DESIGN.md 6.4: serial on
        # one stream a pair is 10 % faster than its two launches at W = 664, equal at W = 1328, 2 - 7 % slower at W = 2656 (real
        # activations); end to end +0.65 % with one batch in flight, -0.4 .. -0.7 % with the default two (the other stream's launches
        # already fill the tail rounds, and finer launches interleave better).
        if not self.h16 or devswitch.get("RD_PAIR", "0") in ("", "0"):
            return
        max_w = int(devswitch.get("RD_PAIR_MAXW", "100000"))
        same_input_only = devswitch.get("RD_PAIR", "0") == "2"    # (round 6 experiment: only the tower convs that read the SAME tensor -- conv_0 of a level)

        def sig(st):
            if st["kind"] != "conv" or not st.get("ex") or not st.get("fold") or st.get("sc") or st.get("s2view") or st.get("x2") is not None:
                return None
            if st["res"] is not None or st["cout"] != 128 or st["stride_w"] != 1 or tuple(st["k"]) != (3, 3) or st["flags"] != RD_RELU_POST:
                return None
            x, o, h = st["x"], st["out"], st.get("head")
            if x.W > max_w:
                return None
            return (x.H, x.W, x.cs, len(st["cmap"]) if st.get("cmap") else st["cin"], o.cs if not h else None,
                    (h["n_off"], h["N"]) if h else None)

        def is_write(key):
            return key in ("out", "out1", "head_out", "head_out1", "keep") or key.startswith("out_")

        out, written, touched, open_ = [], {}, {}, {}
        for st in self.plan.steps:
            k = sig(st)
            if k is not None:
                # the tensor it reads was written before place j, and nothing from place j on touches what it writes
                ready = written.get(st["x"].buf, -1)
                clear = max((touched.get(v.buf, -1) for v in (st["out"], st.get("head_out")) if v is not None), default=-1)
                j = next((j for j in open_.get(k, []) if ready < j and clear < j and (not same_input_only or out[j]["x"] == st["x"])), None)
                if j is not None:
                    open_[k].remove(j)
                    a = out[j]
                    p = dict(kind="conv_pair", name=a["name"] + " + " + st["name"], a=a, b=st, x=a["x"], x1=st["x"], out=a["out"],
                             out1=st["out"])
                    if a.get("head"):
                        p["head_out"], p["head_out1"] = a["head_out"], st["head_out"]
                    out[j] = p
                    for v in (st["out"], st.get("head_out")):
                        if v is not None:
                            written[v.buf] = max(written.get(v.buf, -1), j)
                            touched[v.buf] = max(touched.get(v.buf, -1), j)
                    touched[st["x"].buf] = max(touched.get(st["x"].buf, -1), j)
                    continue
                open_.setdefault(k, []).append(len(out))
            def refs(key, v):   # every buffer reference of a step value, also inside lists / tuples / nested dicts
                if isinstance(v, (TRef, FlatRef)):
                    yield key, v
                    if isinstance(v, TRef) and v.tail is not None:
                        yield key, v.tail
                elif isinstance(v, dict):
                    for k2, v2 in v.items():
                        yield from refs(k2 if isinstance(k2, str) else key, v2)
                elif isinstance(v, (list, tuple)):
                    for v2 in v:
                        yield from refs(key, v2)
            for key, v in st.items():
                for k2, r in refs(key, v):
                    touched[r.buf] = len(out)
                    if is_write(k2):
                        written[r.buf] = len(out)
            out.append(st)
        self.plan.steps = out

    def _fuse_blocks(self):
        """16-bit: a 64-channel BasicBlock at stride 1 (dla_backbone.py:18-56: conv1 3x3 + BN + ReLU -> conv2 3x3 + BN, + shortcut, ReLU)
        whose intermediate tensor has no other reader runs as ONE launch (rd_block64_bn_act, csrc/k_block.h): conv1's result stays in
        LDS as conv2's halo image.  Both shortcut forms: identity (residual = the block input) and the fused 1x1 projection of the
        block input.  Plan step kind "block": a / b = the two conv steps as they were (bit-identical results).
        RD_NO_FUSE_BLOCK=1 keeps the two launches (A/B runs)."""
        if not self.h16 or devswitch.get("RD_NO_FUSE_BLOCK"):
            return
        steps = self.plan.steps
        uses = {}

        def scan(i, key, v):        # every TRef a step holds, also inside nested dicts / lists (fused head_out steps, 'sc' dicts: ADVICE r5)
            if isinstance(v, (TRef, FlatRef)):
                uses.setdefault(v.buf, []).append((i, key))
                if isinstance(v, TRef) and v.tail is not None:
                    scan(i, key + ".tail", v.tail)
            elif isinstance(v, dict):
                for k2, v2 in v.items():
                    scan(i, key + "." + str(k2), v2)
            elif isinstance(v, (list, tuple)):
                for k2, v2 in enumerate(v):
                    scan(i, key + "." + str(k2), v2)
        for i, st in enumerate(steps):
            for key, v in st.items():
                scan(i, key, v)
        # a tensor the graph RETURNS has a reader outside the step list: its buffer must survive
        for o in (self.plan.outputs if isinstance(self.plan.outputs, (list, tuple)) else []):
            scan(-1, "output", o)

        def plain64(st, first=False):
            """a folded 3x3 stride-1 conv to 64 channels from 64 channels -- or, as conv1 of the network's first block, from one
            16-channel granule (the 8-channel range image / KITTI's 5 channels in a zero-padded 16-channel buffer)"""
            if not (st["kind"] == "conv" and tuple(st["k"]) == (3, 3) and st["stride_w"] == 1 and st["cout"] == 64 and st.get("ex") and
                    st.get("fold") and not st.get("head") and st.get("x2") is None and st["x"].tail is None):
                return False
            if first and st["x"].cs - st["x"].co == 16 and st["x"].C <= 16:
                return not devswitch.get("RD_NO_FUSE_FIRST")      # (A/B switch: keep the first block as two launches)
            return st["cin"] == 64 and st["x"].C == 64 and not st.get("cmap")

        out, i = [], 0
        while i < len(steps):
            a = steps[i]
            b = steps[i + 1] if i + 1 < len(steps) else None
            ok = b is not None and plain64(a, first=True) and plain64(b) and a["flags"] == RD_RELU_POST and a["res"] is None and not a.get("sc") and \
                b["flags"] == (RD_ADD | RD_RELU_POST) and b["x"] == a["out"] and a["out"].co == 0 and a["out"].cs == 64 and \
                sorted(uses.get(a["out"].buf, [])) == [(i, "out"), (i + 1, "x")] and b["out"].buf != a["x"].buf
            if ok:
                small = a["x"].C <= 16
                if b.get("sc"):      # projection shortcut of the block input (the same tensor conv1 reads)
                    ok = b["res"] is None and b["sc_x"] == a["x"] and (small or (b["sc"]["cin"] == 64 and not b["sc"].get("cmap")))
                else:                # identity shortcut: the residual IS the block input
                    ok = b["res"] == a["x"] and not small
            if ok:
                blk = dict(kind="block", name=a["name"] + " + " + b["name"], a=a, b=b, x=a["x"], out=b["out"])
                a["in_block"] = b["in_block"] = True
                self.plan.buffers.pop(a["out"].buf, None)      # the intermediate tensor no longer exists
                out.append(blk)
                i += 2
            else:
                out.append(a)
                i += 1
        self.plan.steps = out

    def _fuse_head_out(self):
        """bf16: the last conv of a head tower whose ONLY consumer is one 1x1 output conv (rpn_cls_logit / rpn_reg_delta of a
        single-class head) runs as rd_conv2d_bn_act_head_out: the output conv is applied in the 3x3 kernel's epilogue and the
        128-channel tower output is never written to HBM nor read back.  (Two classes read the tensor twice: left alone.)"""
        if not self.h16 or devswitch.get("RD_NO_FUSE_HEAD"):
            return
        steps = self.plan.steps
        uses = {}
        for i, st in enumerate(steps):
            for key, v in st.items():
                if isinstance(v, TRef):
                    uses.setdefault(v.buf, []).append((i, key))
        drop = set()
        for i, st in enumerate(steps):
            if st["kind"] != "head_out" or st["nout"] > 8:
                continue
            x = st["x"]
            refs = uses.get(x.buf, [])
            prod = [j for j, key in refs if key == "out"]
            if len(refs) != 2 or len(prod) != 1 or x.co != 0 or x.C != 128:
                continue
            c = steps[prod[0]]
            if not (c["kind"] == "conv" and tuple(c["k"]) == (3, 3) and c["cout"] == 128 and c["stride_w"] == 1 and
                    c["flags"] == RD_RELU_POST and c.get("res") is None and c["out"] == x and c.get("x2") is None):
                continue
            c["head"] = dict(name=st["name"], rows=st["rows"], nout=st["nout"], out=st["out"], n_off=st["n_off"], N=st["N"])
            c["head_out"] = st["out"]        # top level: the executor's buffer liveness pass looks at step values
            drop.add(i)
        self.plan.steps = [st for i, st in enumerate(steps) if i not in drop]

    # ---- bookkeeping ---------------------------------------------------------------------------------------
    def _count_consumers(self, root):
        seen = set()

        def walk(s):
            for i in s.inputs:
                self.consumers[i.uid] = self.consumers.get(i.uid, 0) + 1
                if i.uid not in seen:
                    seen.add(i.uid)
                    walk(i)
        walk(root)

    def new_buf(self, nbytes, persistent=False, zero=False):
        self.nbuf += 1
        self.plan.buffers[self.nbuf] = dict(nbytes=int(nbytes), persistent=persistent, zero=zero)
        return self.nbuf

    def new_act(self, C, H, W, persistent=False, zero=False, cs=None):
        cs = cs or -(-C // self.gran) * self.gran
        buf = self.new_buf(self.B * H * W * cs * self.esz, persistent or cs != C, zero or cs != C)
        return TRef(buf, C, H, W, cs, 0)

    def step(self, kind, **kw):
        kw["kind"] = kind
        # A "virtual concat" TRef (tail set: [feature map | variable], never written) describes ONE buffer's channels only; the single
        # consumer that knows how to read both tensors is the 3x3 conv over a two-tensor input (x + x2, rd_conv3x3_bn_act_cat).  Any
        # other step handed such a reference -- transposed conv, 1x1 conv, Meta-Kernel, residual, shortcut input, output conv --
        # would silently read the feature map's channels alone, so it is refused here, once, for every step builder.
        for key, v in kw.items():
            if isinstance(v, TRef) and v.tail is not None and not (kind == "conv" and key == "x" and kw.get("x2") is v.tail and tuple(kw.get("k", ())) == (3, 3)):
                raise NotImplementedError("%s step %r: operand %r is a virtual concat [%d channels | %d more from another tensor]; only a "
                                          "3x3 stride-1 convolution can read it (set RD_CONCAT_BUFFER=1 for a materialised concat)" %
                                          (kind, kw.get("name", ""), key, v.C, v.tail.C))
        self.plan.steps.append(kw)

    def want_input(self, name):
        if name not in self.shapes:
            raise KeyError("lowering needs the shape of input variable %r" % name)
        self.plan.input_vars[name] = tuple(self.shapes[name])
        return tuple(self.shapes[name])

    # ---- activations (4-D) ---------------------------------------------------------------------------------------
    def emit_act(self, s, dest=None):
        s = _strip_cast(s)
        key = s.uid
        if s.op == "var" and dest is not None:  # an input variable may be staged again, straight into a concat buffer
            return self._emit_act(s, dest)
        if key in self.memo:
            if dest is not None:
                raise NotImplementedError("%s is needed both stand-alone and inside a concat buffer" % s.name)
            return self.memo[key]
        r = self._emit_act(s, dest)
        if dest is None:
            self.memo[key] = r
        return r

    def _out(self, C, H, W, dest):
        if dest is None:
            return self.new_act(C, H, W)
        buf, cs, co = dest
        return TRef(buf, C, H, W, cs, co)

    def _emit_act(self, s, dest):
        if s.op == "var":
            C, H, W = self.want_input(s.name)
            out = self._out(C, H, W, dest) if dest else self.new_act(C, H, W, persistent=True)
            pad = (out.cs - out.co - C) if dest is None else 0
            self.step("nchw_in", name=s.name, out=out, zero_pad=pad)
            return out
        if _is(s, "Activation", act_type="relu"):
            inner = s.inputs[0]
            if inner.op == "elemwise_add":
                return self._residual_block(inner, dest)
            if inner.op == "BatchNorm" and inner.inputs[0].op == "Convolution":
                conv = inner.inputs[0]
                meta = self._match_meta(conv)
                if meta is not None:
                    return self._emit_meta(meta, conv, inner, dest)
                return self._conv_bn(conv, inner, RD_RELU_POST, None, dest)
        if s.op == "BatchNorm" and s.inputs[0].op == "Convolution":
            return self._conv_bn(s.inputs[0], s, 0, None, dest)
        if s.op == "elemwise_add":
            return self._agg_add(s, dest)
        if _is(s, "concat", dim=1):
            return self._concat(s, dest)
        raise NotImplementedError("no HIP lowering for %r (op %s) as an activation" % (s.name, s.op))

    def _conv_geom(self, conv, x):
        k = conv.attrs["kernel"]
        st = conv.attrs["stride"]
        pad = conv.attrs["pad"]
        if k not in ((1, 1), (3, 3)) or st[0] != 1 or st[1] not in (1, 2) or pad != ((k[0] - 1) // 2, (k[1] - 1) // 2):
            raise NotImplementedError("Convolution %s: kernel %s stride %s pad %s" % (conv.name, k, st, pad))
        Wout = (x.W + 2 * pad[1] - k[1]) // st[1] + 1
        return k, st[1], Wout

    def _conv_bn(self, conv, bn, flags, residual, dest, sc=None):
        """sc = (conv1x1, bn) of a projection shortcut to fuse into this 3x3 conv's epilogue (bf16), or None."""
        if not conv.attrs["no_bias"]:
            raise NotImplementedError("Convolution %s with bias followed by BatchNorm" % conv.name)
        x = self.emit_act(conv.inputs[0])
        k, sw, Wout = self._conv_geom(conv, x)
        cout_l = conv.attrs["num_filter"]          # logical output channels (BackboneParam.num_filter, dla_backbone.py:59-103,130-161)
        # The conv kernels are instantiated for 64 and 128 output channels.  Any other width up to 128 runs on the next of the two:
        # the packer gets zero weight rows and a zero shift for the padding channels, which therefore come out as exact zeros
        # (relu(0 + 0), also through a residual add of two such tensors), and every consumer reads the logical channels only
        # (x.C of a TRef is logical, its channel stride physical).
        cout = 64 if cout_l <= 64 else 128 if cout_l <= 128 else None
        if cout is None:
            raise NotImplementedError("Convolution %s: %d output channels.  The HIP conv kernels are instantiated for 64 and 128 output "
                                      "channels (other widths up to 128 run zero-padded on those); more than 128 channels per layer "
                                      "(dla_backbone.py num_filter) has no lowering" % (conv.name, cout_l))
        if cout != cout_l and dest is not None:
            raise NotImplementedError("Convolution %s: %d output channels inside a channel concat (only 64 / 128 there)" % (conv.name, cout_l))
        out = self._out(cout, x.H, Wout, dest) if cout == cout_l else self.new_act(cout_l, x.H, Wout, cs=cout)
        # extended 3x3 entry (bf16): stride (1,2) on the pixel-pair view (even width), and / or the fused projection shortcut
        # and, unless RD_NO_FOLD is set, every bf16 3x3 conv with the BatchNorm scale folded into its weights (RD_SCALE_FOLDED)
        fold = self.h16 and k == (3, 3) and not devswitch.get("RD_NO_FOLD")
        s2view = sw == 2 and x.W % 2 == 0 and not devswitch.get("RD_NO_S2_VIEW")
        ex = self.h16 and k == (3, 3) and (sc is not None or s2view or (fold and sw == 1))
        kw_fold = dict(fold=bool(ex and (fold or sc is not None)), s2view=bool(ex and s2view))
        kw = {}
        if sc is not None:
            scx = self.emit_act(sc[0].inputs[0])
            kw["sc"] = dict(name=sc[0].name, bn=sc[1].name, eps=sc[1].attrs["eps"], cin=scx.C, cmap=scx.cmap)
            kw["sc_x"] = scx                      # top level: the executor's buffer liveness pass looks at step values
        if x.tail is not None:
            kw["x2"] = x.tail                     # (top level: buffer liveness)
        self.step("conv", name=conv.name, bn=bn.name, eps=bn.attrs["eps"], x=x, out=out, res=residual,
                  cin=sum(1 for m in x.cmap if m >= 0) if x.tail is not None else x.C,   # (logical channels of a virtual concat)
                  cout=cout, cout_logical=cout_l, k=k, stride_w=sw, flags=flags, cmap=x.cmap, ex=ex, **kw_fold, **kw)
        return out

    def _fusable_projection(self, main_conv, sc):
        """sc = BN(Convolution 1x1, no bias) with the main 3x3 conv's stride, at most 128 input channels, and (stride 2) an even
        input width: the shortcut the persistent 3x3 kernel can accumulate in its epilogue (bf16 only)."""
        if not self.h16 or devswitch.get("RD_NO_FUSE_SC"):
            return None
        if not (sc.op == "BatchNorm" and sc.inputs[0].op == "Convolution"):
            return None
        c = sc.inputs[0]
        if c.attrs["kernel"] != (1, 1) or not c.attrs["no_bias"] or c.attrs["stride"] != main_conv.attrs["stride"] or \
                c.attrs["num_filter"] != main_conv.attrs["num_filter"] or self.consumers.get(sc.uid, 0) != 1:
            return None
        x0 = self.emit_act(c.inputs[0])
        if x0.C > 128 or (c.attrs["stride"][1] == 2 and x0.W % 2) or c.attrs["stride"][1] not in (1, 2):
            return None
        return (c, sc)

    def _residual_block(self, add, dest):
        a, b = add.inputs
        main = sc = None
        for m, o in ((a, b), (b, a)):
            if m.op == "BatchNorm" and m.inputs[0].op == "Convolution" and m.inputs[0].attrs["kernel"] == (3, 3):
                main, sc = m, o
                break
        if main is None:
            raise NotImplementedError("relu(add) without a BN(conv3x3) branch at %s" % add.name)
        proj = self._fusable_projection(main.inputs[0], sc)
        if proj is not None:
            return self._conv_bn(main.inputs[0], main, RD_ADD | RD_RELU_POST, None, dest, sc=proj)
        res = self.emit_act(sc)
        return self._conv_bn(main.inputs[0], main, RD_ADD | RD_RELU_POST, res, dest)

    def _agg_add(self, add, dest):
        a, b = add.inputs
        for skip, up in ((a, b), (b, a)):
            if _is(up, "Activation", act_type="relu") and up.inputs[0].op == "BatchNorm" and \
                    up.inputs[0].inputs[0].op == "Deconvolution":
                bn = up.inputs[0]
                dc = bn.inputs[0]
                x = self.emit_act(dc.inputs[0])
                res = self.emit_act(skip)
                kh, kw = dc.attrs["kernel"]
                sh, sw = dc.attrs["stride"]
                ph, pw = dc.attrs["pad"]
                if kh != 3 or sh != 1 or ph != 1:
                    raise NotImplementedError("Deconvolution %s: kernel/stride/pad height" % dc.name)
                Wout = (x.W - 1) * sw - 2 * pw + kw
                cout_l = dc.attrs["num_filter"]
                cout = 64 if cout_l <= 64 else 128 if cout_l <= 128 else None        # (zero-padded to a kernel width like _conv_bn)
                if cout is None or (cout != cout_l and dest is not None):
                    raise NotImplementedError("Deconvolution %s: %d output channels (at most 128; 64 / 128 inside a concat)" % (dc.name, cout_l))
                out = self._out(cout, x.H, Wout, dest) if cout == cout_l else self.new_act(cout_l, x.H, Wout, cs=cout)
                if (res.C, res.H, res.W) != (cout_l, x.H, Wout) or (cout != cout_l and (res.cs != cout or res.co != 0)):
                    raise ValueError("agg add shape mismatch at %s" % add.name)
                fold = self.h16 and cout in (64, 128) and not devswitch.get("RD_NO_FOLD")
                # every phase a 3 x 2 tap set (kw = 2 sw, pad = sw / 2: k(3,8) s4 p2, k(3,4) s2 p1): all phases in ONE launch
                # (rd_deconv2d_bn_act_all; the executor checks this against rd_deconv2d_all_phases_ok)
                one = bool(fold and kw == 2 * sw and 2 * pw == sw and not devswitch.get("RD_DECONV_PER_PHASE"))
                self.step("deconv", name=dc.name, bn=bn.name, eps=bn.attrs["eps"], x=x, out=out, res=res, cin=x.C,
                          cout=cout, cout_logical=cout_l, k=(kh, kw), stride_w=sw, pad_w=pw, flags=RD_RELU_PRE | RD_ADD, fold=fold, one_launch=one)
                return out
        raise NotImplementedError("elemwise_add %s is not skip + relu(BN(Deconvolution))" % add.name)

    def _concat(self, s, dest):
        if dest is not None:
            raise NotImplementedError("nested channel concat")
        parts = [_strip_cast(i) for i in s.inputs]
        dims = []
        for p in parts:
            if p.op == "var":
                dims.append(self.want_input(p.name))
            else:
                dims.append(None)
        # emit non-var parts first to learn their shape, writing into the shared buffer
        # channel counts: var from its shape, computed parts from their producing conv
        def chans(p):
            if p.op == "var":
                return self.shapes[p.name][0]
            q = p
            while q.op in ("Activation", "BatchNorm", "elemwise_add", "cast"):
                q = q.inputs[0]
            return q.attrs["num_filter"]
        cs_list = [chans(p) for p in parts]
        Ctot = sum(cs_list)
        # 16-bit, [input variable(s) of at most one channel granule in total | ONE computed feature map of a multiple of 32 channels]
        # (dla_backbone.py:153-154: the 8-channel range image + agg3): no shared buffer.  The feature map keeps its own buffer
        # (128-byte pixel rows instead of rows at a 160-byte pitch that straddle cache lines), the variable its zero-padded
        # one-granule buffer, and the consuming convs read [feature map | variable] (rd_conv3x3_bn_act_cat) with their weight
        # columns permuted accordingly (cmap).  RD_CONCAT_BUFFER=1: the shared buffer, for A/B runs.
        acts = [i for i, p in enumerate(parts) if p.op != "var"]
        if self.h16 and len(parts) == 2 and len(acts) == 1 and cs_list[acts[0]] % 32 == 0 and cs_list[1 - acts[0]] <= self.gran and \
                not devswitch.get("RD_CONCAT_BUFFER") and not devswitch.get("RD_NO_FOLD") and devswitch.get("RD_PAIR", "0") in ("", "0"):
            # (RD_PAIR=1, the opt-in two-problem tower launches, keeps the shared buffer: that launch form takes one input tensor)
            ia, iv = acts[0], 1 - acts[0]
            fa = self.emit_act(parts[ia])
            fv = self.emit_act(parts[iv])
            if (fa.H, fa.W) != (fv.H, fv.W) or fa.cmap is not None or fa.tail is not None or fa.co != 0 or fv.co != 0:
                raise ValueError("concat %s: parts do not line up" % s.name)
            start = [0, cs_list[0]]               # logical channel offset of each part in the concatenation
            cmap = list(range(start[ia], start[ia] + fa.C)) + list(range(start[iv], start[iv] + fv.C)) + [-1] * (fv.cs - fv.C)
            return TRef(fa.buf, fa.C, fa.H, fa.W, fa.cs, fa.co, tuple(cmap), fv)
        ref_hw = None
        for p, d in zip(parts, dims):
            if d is not None:
                ref_hw = (d[1], d[2])
        if ref_hw is None:
            raise NotImplementedError("concat without a variable input: spatial size unknown before emission")
        # every part must start on a 16-byte slot boundary (aligned stores of its producer): parts that do not are
        # followed by zero channels, and the consumer conv gets zero weight columns there (cmap)
        offs, cmap, co = [], [], 0
        align = max(1, self.gran // 2)
        for c in cs_list:
            pad = -co % align
            cmap += [-1] * pad
            co += pad
            offs.append(co)
            cmap += list(range(len([m for m in cmap if m >= 0]), len([m for m in cmap if m >= 0]) + c))
            co += c
        Cphys = co
        out = self.new_act(Cphys, ref_hw[0], ref_hw[1], persistent=True, zero=True)
        if Cphys != Ctot:
            out.cmap = tuple(cmap)
        for p, c, o in zip(parts, cs_list, offs):
            r = self.emit_act(p, dest=(out.buf, out.cs, o))
            if (r.H, r.W) != ref_hw:
                raise ValueError("concat %s: spatial mismatch" % s.name)
        return out

    # ---- Meta-Kernel unit ------------------------------------------------------------------------------------------
    def _match_meta(self, agg_conv):
        """agg_conv = Convolution 1x1 whose input is relu(BN(reshape(multiply(reshape(im2col(data)), mlp(...)))))."""
        try:
            a = agg_conv.inputs[0]
            if not _is(a, "Activation", act_type="relu"):
                return None
            bn1 = a.inputs[0]
            if bn1.op != "BatchNorm" or bn1.inputs[0].op != "reshape":
                return None
            mul = bn1.inputs[0].inputs[0]
            if mul.op != "multiply":
                return None
        except (IndexError, AttributeError):
            return None
        ds, wts = mul.inputs
        if not (ds.op == "reshape" and ds.inputs[0].op == "im2col"):
            raise NotImplementedError("meta kernel: unexpected data-sample branch at %s" % mul.name)
        im_d = ds.inputs[0]
        data = im_d.inputs[0]
        if im_d.attrs != dict(kernel=(3, 3), stride=(1, 1), dilate=(1, 1), pad=(1, 1)):
            raise NotImplementedError("meta kernel: only 3x3/stride 1/pad 1 sampling")
        # weights branch: reshape(conv(relu(conv(reshape(broadcast_minus(reshape(im2col(coord)), expand_dims(coord)))))))
        c1 = wts.inputs[0] if wts.op == "reshape" else None
        ok = c1 is not None and c1.op == "Convolution" and _is(c1.inputs[0], "Activation", act_type="relu")
        c0 = c1.inputs[0].inputs[0] if ok else None
        ok = ok and c0.op == "Convolution" and c0.inputs[0].op == "reshape" and c0.inputs[0].inputs[0].op == "broadcast_minus"
        if not ok:
            raise NotImplementedError("meta kernel: unexpected MLP branch (only fc+relu+fc without norm)")
        bm = c0.inputs[0].inputs[0]
        nb, ctr = bm.inputs
        if not (nb.op == "reshape" and nb.inputs[0].op == "im2col" and ctr.op == "expand_dims"):
            raise NotImplementedError("meta kernel: unexpected relative-coordinate branch")
        coord = _strip_cast(ctr.inputs[0])
        if _strip_cast(nb.inputs[0].inputs[0]).uid != coord.uid or coord.op != "var":
            raise NotImplementedError("meta kernel: coordinates must be one input variable")
        if (c0.attrs["num_filter"], c1.attrs["num_filter"], agg_conv.attrs["num_filter"]) != (32, 64, 64) or \
                c0.attrs["no_bias"] or c1.attrs["no_bias"] or not agg_conv.attrs["no_bias"]:
            raise NotImplementedError("meta kernel %s: MLP %d -> %d, aggregation to %d channels.  The fused Meta-Kernel is built for the shipped "
                                      "unit only: 3 -> 32 -> 64 MLP with biases (meta_kernel_units / fc channels of the reference config), "
                                      "64 data channels, 9 x 64 = 576 -> 64 aggregation without bias" %
                                      (agg_conv.name, c0.attrs["num_filter"], c1.attrs["num_filter"], agg_conv.attrs["num_filter"]))
        return dict(data=data, coord=coord.name, mlp0=c0.name, mlp1=c1.name, bn1=bn1.name, eps1=bn1.attrs["eps"])

    def _emit_meta(self, m, agg_conv, bn2, dest):
        x = self.emit_act(m["data"])
        if x.C != 64:
            raise NotImplementedError("meta kernel %s: %d data channels (the fused Meta-Kernel is built for 64)" % (agg_conv.name, x.C))
        cshape = self.want_input(m["coord"])
        if cshape != (3, x.H, x.W):
            raise ValueError("meta kernel: coord shape %s vs data %s" % (cshape, (x.H, x.W)))
        out = self._out(64, x.H, x.W, dest)
        self.step("meta", x=x, out=out, coord=m["coord"], mlp0=m["mlp0"], mlp1=m["mlp1"], bn1=m["bn1"], eps1=m["eps1"],
                  agg=agg_conv.name, bn2=bn2.name, eps2=bn2.attrs["eps"])
        return out

    # ---- head / post-processing values -----------------------------------------------------------------------------
    def emit_value(self, s):
        s0 = s
        key = ("v", s.uid, s.index)
        if key in self.memo:
            return self.memo[key]
        if s.op == "var":
            self.plan.input_vars.setdefault(s.name, None)
            v = ("input", s.name)
        elif s.op == "zeros":
            v = ("zeros", s.attrs["shape"])
        elif s.op == "Custom" and s.attrs["op_type"] == "get_sorted_foreground":
            v = ("flat", self._sorted_fg(s)[s.index])
        elif s.op == "Custom" and s.attrs["op_type"] == "batch_rotated_iou":
            v = ("flat", self._batch_riou(s))
        elif s.op == "Decode3DBbox":
            v = ("flat", self._decode(s))
        elif s.op == "NMS3D":
            keep, final = self._nms3d(s)
            v = ("flat_i32", keep) if s.index == 0 else ("flat", final)
        else:
            raise NotImplementedError("graph output %r (op %s) has no HIP lowering" % (s0.name, s.op))
        self.memo[key] = v
        return v

    def _flat_levels(self, s, transposed):
        """s = concat over levels of squeeze(slice_axis(reshape(cast(Convolution 1x1 + bias)))) [transpose for deltas].
        Returns [(conv, class index)] in concat order."""
        levels = s.inputs if s.op == "concat" else [s]
        out = []
        for lv in levels:
            q = lv
            if transposed:
                if not _is(q, "transpose", axes=(0, 2, 1)):
                    raise NotImplementedError("bbox_delta level is not transposed (0,2,1)")
                q = q.inputs[0]
            if q.op != "squeeze" or q.inputs[0].op != "slice_axis" or q.inputs[0].inputs[0].op != "reshape":
                raise NotImplementedError("per-class flatten chain not recognised at %s" % lv.name)
            sl = q.inputs[0]
            cls_i = sl.attrs["begin"]
            if sl.attrs["axis"] != 1 or sl.attrs["end"] != cls_i + 1:
                raise NotImplementedError("slice_axis %s" % sl.name)
            conv = _strip_cast(sl.inputs[0].inputs[0])
            if conv.op != "Convolution" or conv.attrs["kernel"] != (1, 1) or conv.attrs["no_bias"]:
                raise NotImplementedError("head output must be a 1x1 Convolution with bias at %s" % lv.name)
            out.append((conv, cls_i))
        return out

    def _sorted_fg(self, s):
        key = ("sfg", s.uid)
        if key in self.memo:
            return self.memo[key]
        score, delta, pc, mask = s.inputs
        apply_sigmoid = 0
        if _is(score, "Activation", act_type="sigmoid"):
            apply_sigmoid = 1
            score = score.inputs[0]
        lv_s = self._flat_levels(score, False)
        lv_d = self._flat_levels(delta, True)
        if len(lv_s) != len(lv_d):
            raise ValueError("score / delta level count")
        feats, N = [], 0
        for (cs_, ci), (cd_, di) in zip(lv_s, lv_d):
            fs = self.emit_act(cs_.inputs[0])
            fd = self.emit_act(cd_.inputs[0])
            feats.append((cs_, ci, fs, cd_, di, fd, N))
            N += fs.H * fs.W
        ncls = lv_s[0][0].attrs["num_filter"]
        D = lv_d[0][0].attrs["num_filter"] // ncls
        logit = FlatRef(self.new_buf(self.B * N * 4, persistent=True), (N,))
        dl = FlatRef(self.new_buf(self.B * N * D * 4, persistent=True), (N, D))
        for cs_, ci, fs, cd_, di, fd, off in feats:
            self.step("head_out", name=cs_.name, x=fs, out=logit, n_off=off, N=N, rows=(ci, ci + 1), nout=1)
            self.step("head_out", name=cd_.name, x=fd, out=dl, n_off=off, N=N, rows=(di * D, di * D + D), nout=D)
        pcs = self._concat_vars(pc, 3)
        msk = self._concat_vars(mask, None)
        k = int(s.attrs["num_fgs"])
        if N < k:
            raise ValueError("get_sorted_foreground: N (%d) < num_fgs (%d)" % (N, k))  # get_sorted_foreground.py:65
        o_s = FlatRef(self.new_buf(self.B * k * 4, persistent=True), (k,))
        o_d = FlatRef(self.new_buf(self.B * k * D * 4), (k, D))
        o_p = FlatRef(self.new_buf(self.B * k * 3 * 4), (k, 3))
        self.step("sorted_fg", score=logit, delta=dl, pc=pcs, mask=msk, N=N, k=k, D=D, apply_sigmoid=apply_sigmoid,
                  out_score=o_s, out_delta=o_d, out_pc=o_p)
        self.memo[key] = (o_s, o_d, o_p)
        return self.memo[key]

    def _concat_vars(self, s, last):
        parts = s.inputs if s.op == "concat" else [s]
        names, n = [], 0
        for p in parts:
            p = _strip_cast(p)
            if p.op != "var":
                raise NotImplementedError("pc / mask inputs must be input variables")
            shp = self.want_input(p.name)
            names.append((p.name, shp[0]))
            n += shp[0]
        ref = FlatRef(self.new_buf(self.B * n * (last or 1) * 4, persistent=True), (n, last) if last else (n,))
        self.step("concat_in", names=names, out=ref, last=last or 1, N=n)
        return ref

    def _decode(self, s):
        d, p = s.inputs
        if not (d.op == "Custom" and p.op == "Custom" and d.uid == p.uid and d.index == 1 and p.index == 2):
            raise NotImplementedError("Decode3DBbox inputs must be outputs 1,2 of get_sorted_foreground")
        o_s, o_d, o_p = self._sorted_fg(d)
        k = o_d.shape[0]
        out = FlatRef(self.new_buf(self.B * k * 10 * 4, persistent=True), (k, 10))
        self.step("decode", delta=o_d, pc=o_p, out=out, k=k, box_type=o_d.shape[1], is_bin=int(s.attrs["is_bin"]))
        return out


    def _batch_riou(self, s):
        """Custom op 'batch_rotated_iou' (operator_py/batch_rotated_iou.py): proposal = a (B,N,10) box tensor of this graph,
        gt_bbox = an input variable (B, n_gt, 8) -> iou_map (B,N)."""
        key = ("briou", s.uid)
        if key in self.memo:
            return self.memo[key]
        iou_type = s.attrs.get("iou_type", "bev")
        if iou_type not in ("bev", "3d"):
            raise ValueError("batch_rotated_iou: iou_type %r ('bev' or '3d')" % iou_type)           # batch_rotated_iou.py:40-41,91-92
        gdim = 8 if iou_type == "bev" else 7                                                      # :86-89
        prop, gt = s.inputs
        kind, boxes = self.emit_value(prop)
        gt = _strip_cast(gt)
        if kind != "flat" or len(boxes.shape) != 2 or boxes.shape[1] != 10 or gt.op != "var":
            raise NotImplementedError("batch_rotated_iou: proposal must be a (B,N,10) box tensor, gt_bbox an input variable")
        gshape = self.want_input(gt.name)
        if len(gshape) != 2 or gshape[1] != gdim or gshape[0] > 256:
            raise ValueError("batch_rotated_iou (%s): gt_bbox shape %s (need (n_gt <= 256, %d))" % (iou_type, gshape, gdim))   # batch_rotated_iou.py:78-89
        out = FlatRef(self.new_buf(self.B * boxes.shape[0] * 4, persistent=True), (boxes.shape[0],))
        self.step("batch_riou", boxes=boxes, gt=gt.name, N=boxes.shape[0], n_gt=gshape[0], out=out, iou_type=iou_type)
        self.memo[key] = out
        return out

    def _nms3d(self, s):
        key = ("nms3d", s.uid)
        if key in self.memo:
            return self.memo[key]
        (bx,) = s.inputs
        if bx.op != "Decode3DBbox":
            raise NotImplementedError("NMS3D input must be the Decode3DBbox output (score-sorted boxes)")
        boxes = self._decode(bx)
        n, mk = boxes.shape[0], int(s.attrs["max_keep"])
        keep = FlatRef(self.new_buf(self.B * mk * 4, persistent=True), (mk,))
        final = FlatRef(self.new_buf(self.B * mk * 10 * 4, persistent=True), (mk, 10))
        self.step("nms3d", boxes=boxes, N=n, thr=float(s.attrs["iou_thres"]), max_keep=mk,
                  normal_iou=int(s.attrs["normal_iou"]), keep=keep, out=final)
        self.memo[key] = (keep, final)
        return self.memo[key]


def lower(test_symbol, input_shapes, dtype=RD_BF16, batch=1):
    """test_symbol: the Group returned by RangeRCNN.get_test_symbol.  Returns a Plan."""
    return Lowering(test_symbol, input_shapes, dtype, batch).plan
