"""The ``mxnext.simple`` names the RangeDet test graph uses, re-expressed over the recording IR (rangedet_amd.mx).

Reference: mxnext/simple.py -- conv :123-158 (defaults no_bias=True, pad=((k-1)*d+1)//2), deconv :545-580,
relu :32-50, sigmoid :53-68, aliases :449-483, convnormrelu :502-508.  Signatures and defaults are kept so the
reference's model code runs unchanged against this module; initialisers and lr/wd multipliers are training-only and
accepted but ignored.
"""
import numpy as np

from .. import mx

__all__ = ["var", "relu", "sigmoid", "conv", "deconv", "add", "reshape", "transpose", "concat", "group", "to_fp16",
           "to_fp32", "convnormrelu", "gauss", "identity"]


def _two(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _default_pad(kernel, dilate, pad, name):
    if pad is None:
        if kernel[0] % 2 != 1:
            raise AssertionError("Specify pad for an even kernel size for {}".format(name))
        pad = ((kernel[0] - 1) * dilate[0] + 1) // 2
    return _two(pad)


def var(name, **kwargs):
    return mx.var(name, **kwargs)


def _derived(data, suffix):
    base = data.name
    if base.endswith("_bn") or base.endswith("_gn"):
        base = base[:-3]
    return base + suffix


def relu(data, name=None, act_type="relu"):
    if act_type != "relu":
        raise NotImplementedError("relu6 is not on the RangeDet test path")
    return mx.Activation(data, name=name or _derived(data, "_relu"), act_type="relu")


def sigmoid(data, name=None):
    return mx.Activation(data, name=name or _derived(data, "_sigmoid"), act_type="sigmoid")


def conv(data, name, filter, kernel=1, stride=1, pad=None, dilate=1, num_group=1, no_bias=True, init=None,
         lr_mult=1.0, wd_mult=1.0, weight=None, bias=None):
    kernel, stride, dilate = _two(kernel), _two(stride), _two(dilate)
    return mx.Convolution(data=data, name=name, weight=weight, bias=bias, num_filter=filter, kernel=kernel,
                          stride=stride, pad=_default_pad(kernel, dilate, pad, name), dilate=dilate,
                          num_group=num_group, workspace=512, no_bias=no_bias)


def deconv(data, name, filter, kernel=1, stride=1, pad=None, dilate=1, num_group=1, no_bias=True, init=None,
           lr_mult=1.0, wd_mult=1.0, weight=None, bias=None):
    kernel, stride, dilate = _two(kernel), _two(stride), _two(dilate)
    return mx.Deconvolution(data=data, name=name, weight=weight, bias=bias, num_filter=filter, kernel=kernel,
                            stride=stride, pad=_default_pad(kernel, dilate, pad, name), dilate=dilate,
                            num_group=num_group, workspace=512, no_bias=no_bias)


def convnormrelu(norm, data, name, filter, kernel=1, stride=1, pad=None, dilate=1, num_group=1, no_bias=True,
                 init=None, conv_lr_mult=1.0, conv_wd_mult=1.0, norm_lr_mult=1.0, norm_wd_mult=1.0):
    c = conv(data, name, filter, kernel, stride, pad, dilate, num_group, no_bias)
    return relu(norm(c, name=name + "_bn"), name + "_relu")


add = mx.elemwise_add
reshape = mx.reshape
identity = lambda data, name=None: data  # noqa: E731


def transpose(data, axes=None, name=None):
    return mx.transpose(data, axes=axes, name=name)


def concat(inputs, name, axis=1):
    assert isinstance(inputs, list), "Concat accepts a list of symbols"
    return inputs[0] if len(inputs) == 1 else mx.concat(*inputs, name=name, dim=axis)


def group(symbols):
    return mx.Group(symbols)


def to_fp16(data, name):
    return mx.cast(data, dtype=np.float16, name=name)


def to_fp32(data, name):
    return mx.cast(data, dtype=np.float32, name=name)


def gauss(std):
    return None  # training-time initialiser; the test path loads parameters
