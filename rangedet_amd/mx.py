"""A recording stand-in for the slice of ``mxnet.symbol`` the RangeDet *test* graph uses.

The reference's model code (rangedet/symbol/**, mxnext/**) builds an MXNet symbol graph; here the same calls build a
light IR (``Symbol`` nodes) that ``rangedet_amd.lower`` turns into a plan of fused HIP kernel launches.  Only the ops
on the inference path exist: var, Convolution, Deconvolution, BatchNorm, Activation,
elemwise_add, cast, im2col, reshape, expand_dims, broadcast_minus, multiply, concat, slice_axis, squeeze, transpose,
Custom('get_sorted_foreground' | 'batch_rotated_iou'), contrib.Decode3DBbox, contrib.NMS3D, zeros, Group.
Anything else raises NotImplementedError at graph-construction time (loudly, not at run time).
"""
import itertools

import numpy as np

_uid = itertools.count()


class Symbol:
    __slots__ = ("op", "inputs", "attrs", "name", "index", "nout", "uid")

    def __init__(self, op, inputs=(), attrs=None, name=None, index=0, nout=1):
        self.op = op
        self.inputs = list(inputs)
        self.attrs = dict(attrs or {})
        self.uid = next(_uid)
        self.name = name if name is not None else "%s%d" % (op.lower(), self.uid)
        self.index = index
        self.nout = nout

    def __mul__(self, other):
        return Symbol("multiply", [self, other])

    def __add__(self, other):
        return Symbol("elemwise_add", [self, other])

    def __iter__(self):
        if self.nout == 1:
            raise TypeError("single-output symbol is not iterable")
        return iter([self[i] for i in range(self.nout)])

    def __getitem__(self, i):
        if self.op == "Group":
            return self.inputs[i]
        if not 0 <= i < self.nout:
            raise IndexError(i)
        s = Symbol.__new__(Symbol)
        s.op, s.inputs, s.attrs, s.uid, s.name, s.index, s.nout = self.op, self.inputs, self.attrs, self.uid, self.name, i, self.nout
        return s

    def __repr__(self):
        return "<Symbol %s %s>" % (self.op, self.name)

    def list_arguments(self):
        seen, out = set(), []

        def walk(s):
            if s.uid in seen:
                return
            seen.add(s.uid)
            for i in s.inputs:
                walk(i)
            if s.op == "var":
                out.append(s.name)
        walk(self)
        return out


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def var(name, **kwargs):
    return Symbol("var", [], kwargs, name)


Variable = var


def _param(name, suffix):
    return Symbol("var", [], {"param": True}, name + suffix)


def Convolution(data, name=None, weight=None, bias=None, num_filter=None, kernel=(1, 1), stride=(1, 1), pad=(0, 0),
                dilate=(1, 1), num_group=1, workspace=512, no_bias=False, **kw):
    if num_group != 1 or _pair(dilate) != (1, 1):
        raise NotImplementedError("Convolution: num_group/dilate other than 1 are not on the RangeDet test path")
    ins = [data, weight if isinstance(weight, Symbol) else _param(name, "_weight")]
    if not no_bias:
        ins.append(bias if isinstance(bias, Symbol) else _param(name, "_bias"))
    return Symbol("Convolution", ins, dict(num_filter=num_filter, kernel=_pair(kernel), stride=_pair(stride),
                                           pad=_pair(pad), no_bias=bool(no_bias)), name)


def Deconvolution(data, name=None, weight=None, bias=None, num_filter=None, kernel=(1, 1), stride=(1, 1), pad=(0, 0),
                  dilate=(1, 1), num_group=1, workspace=512, no_bias=True, **kw):
    if num_group != 1 or _pair(dilate) != (1, 1) or not no_bias:
        raise NotImplementedError("Deconvolution: only no_bias, num_group 1, dilate 1")
    ins = [data, weight if isinstance(weight, Symbol) else _param(name, "_weight")]
    return Symbol("Deconvolution", ins, dict(num_filter=num_filter, kernel=_pair(kernel), stride=_pair(stride),
                                             pad=_pair(pad)), name)


def BatchNorm(data, gamma=None, beta=None, moving_mean=None, moving_var=None, name=None, eps=1e-3, fix_gamma=True,
              use_global_stats=False, momentum=0.9, **kw):
    ins = [data] + [p if isinstance(p, Symbol) else _param(name, s) for p, s in
                    ((gamma, "_gamma"), (beta, "_beta"), (moving_mean, "_moving_mean"), (moving_var, "_moving_var"))]
    return Symbol("BatchNorm", ins, dict(eps=eps, fix_gamma=bool(fix_gamma)), name)


def Activation(data, name=None, act_type="relu"):
    if act_type not in ("relu", "sigmoid"):
        raise NotImplementedError("Activation %s" % act_type)
    return Symbol("Activation", [data], dict(act_type=act_type), name)


def elemwise_add(lhs, rhs, name=None):
    return Symbol("elemwise_add", [lhs, rhs], {}, name)


def cast(data, dtype=None, name=None):
    return Symbol("cast", [data], dict(dtype=np.dtype(dtype).name), name)


def im2col(data, name=None, kernel=(1, 1), stride=(1, 1), dilate=(1, 1), pad=(0, 0)):
    return Symbol("im2col", [data], dict(kernel=_pair(kernel), stride=_pair(stride), dilate=_pair(dilate), pad=_pair(pad)), name)


def reshape(data, shape=None, name=None, **kw):
    return Symbol("reshape", [data], dict(shape=tuple(shape)), name)


def expand_dims(data, axis=None, name=None):
    return Symbol("expand_dims", [data], dict(axis=axis), name)


def broadcast_minus(lhs, rhs, name=None):
    return Symbol("broadcast_minus", [lhs, rhs], {}, name)


def concat(*data, dim=1, name=None):
    return Symbol("concat", list(data), dict(dim=dim), name)


def slice_axis(data, axis=None, begin=None, end=None, name=None):
    return Symbol("slice_axis", [data], dict(axis=axis, begin=begin, end=end), name)


def squeeze(data, axis=None, name=None):
    return Symbol("squeeze", [data], dict(axis=axis), name)


def transpose(data, axes=None, name=None):
    return Symbol("transpose", [data], dict(axes=tuple(axes)), name)


def zeros(shape=None, dtype="float32", name=None):
    return Symbol("zeros", [], dict(shape=tuple(shape)), name)


def Group(symbols):
    return Symbol("Group", list(symbols), {}, "group", nout=len(symbols))


_CUSTOM_OUTPUTS = {"get_sorted_foreground": 3, "batch_rotated_iou": 1}


def Custom(*args, op_type=None, name=None, **kwargs):
    """mx.sym.Custom: tensor inputs are passed by keyword in the reference (builder.py:512-521); their order follows
    the op's list_arguments (get_sorted_foreground.py:52-53, batch_rotated_iou.py:62-63)."""
    if op_type not in _CUSTOM_OUTPUTS:
        raise NotImplementedError("Custom op_type %r" % op_type)
    order = {"get_sorted_foreground": ["cls_score", "bbox_delta", "pc", "mask"],
             "batch_rotated_iou": ["proposal", "gt_bbox"]}[op_type]
    ins = list(args) + [kwargs.pop(k) for k in order if k in kwargs]
    attrs = dict(op_type=op_type, **{k: v for k, v in kwargs.items() if not isinstance(v, Symbol)})
    return Symbol("Custom", ins, attrs, name, nout=_CUSTOM_OUTPUTS[op_type])


class _Contrib:
    @staticmethod
    def Decode3DBbox(bbox_deltas, pc_laser_frame, is_bin=False, name=None):
        return Symbol("Decode3DBbox", [bbox_deltas, pc_laser_frame], dict(is_bin=bool(is_bin)), name)

    @staticmethod
    def NMS3D(boxes, iou_thres, max_keep, normal_iou=False, name=None):
        """_contrib_NMS3D (operator_cxx/contrib/nms_3d.cc:22-68): outputs (idx (B,max_keep) int32, bbox_after_nms
        (B,max_keep,10)); the wnms=False branch of the head (builder.py:530-534)."""
        return Symbol("NMS3D", [boxes], dict(iou_thres=float(iou_thres), max_keep=int(max_keep), normal_iou=bool(normal_iou)),
                      name, nout=2)


contrib = _Contrib()


class _SymNamespace:
    """So that reference-style ``mx.sym.X`` / ``mx.symbol.X`` attribute access works on this module."""
    def __getattr__(self, k):
        g = globals()
        if k in g:
            return g[k]
        raise NotImplementedError("mx.sym.%s is not on the RangeDet test path" % k)


sym = _SymNamespace()
symbol = sym
Symbol_ = Symbol
