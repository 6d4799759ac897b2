"""``MetaKernel`` with the reference's constructor and method surface
(rangedet/symbol/backbone/meta_kernel.py:7-13 ctor, :16-38 sampler_im2col, :40-74 sample_data/sample_coord,
:76-103 relative_coord, :105-164 mlp, :166-240 meta_baseline_bias), recording rangedet_amd.mx ops.

The recorded sub-graph is the reference's un-fused formulation; ``rangedet_amd.lower`` recognises it together with the
five ops DLABackboneBuilder.meta_kernel_conv appends (dla_backbone.py:92-97) and lowers the whole unit to ONE fused HIP
kernel (csrc/k_meta.h).
"""
from __future__ import division

from ... import mx
from ...mxnext.simple import conv, relu, to_fp16


def _two(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class MetaKernel(object):
    def __init__(self, num_batch, feat_height, feat_width, fp16, num_frame=1):
        self.num_batch, self.H, self.W = num_batch, feat_height, feat_width
        self.fp16, self.num_frame = fp16, num_frame

    @staticmethod
    def sampler_im2col(data, name, kernel=1, stride=1, pad=None, dilate=1):
        kernel, stride, dilate = _two(kernel), _two(stride), _two(dilate)
        if pad is None:
            assert kernel[0] % 2 == 1, "Specify pad for an even kernel size for {}".format(name)
            pad = ((kernel[0] - 1) * dilate[0] + 1) // 2
        return mx.im2col(data=data, name=name + "sampler", kernel=kernel, stride=stride, dilate=dilate, pad=_two(pad))

    def _sample(self, tag, tensor, kernel_size):
        return self.sampler_im2col(data=tensor, name=tag, kernel=kernel_size, stride=1, pad=1, dilate=1)

    def sample_data(self, name, data, kernel_size):
        return self._sample(name + "data_", data, kernel_size)

    def sample_coord(self, name, coord, kernel_size):
        return self._sample(name + "coord_", coord, kernel_size)

    def relative_coord(self, sample_coord, center_coord, num_channel_in, kernel_size):
        taps = kernel_size * kernel_size
        neighbours = mx.reshape(sample_coord, shape=(self.num_batch, num_channel_in, taps, self.H, self.W))
        return mx.broadcast_minus(neighbours, mx.expand_dims(center_coord, axis=2), name="relative_dis")

    def mlp(self, data, name, in_channels, norm, channel_list=None, b_mul=1, no_bias=True, use_norm=False):
        assert isinstance(channel_list, list)
        x = mx.reshape(data, shape=(self.num_batch * b_mul, in_channels, -1, self.W))
        last = len(channel_list) - 1
        for i, width in enumerate(channel_list):
            x = conv(x, name=name + "{}_mlp{}".format(self.W, i), filter=width, kernel=1, stride=1, pad=0, dilate=1,
                     no_bias=no_bias)
            if i != last:
                if use_norm:
                    x = norm(x, name=name + "{}_mlp_bn{}".format(self.W, i))
                x = relu(x, name + "{}_mlp_relu{}".format(self.W, i))
        return mx.reshape(x, shape=(self.num_batch * b_mul, channel_list[-1], -1, self.H, self.W))

    def meta_baseline_bias(self, name, data, coord_data, data_channels, coord_channels, channel_list, norm,
                           conv1_filter, kernel_size=3, **kwargs):
        if self.fp16:
            coord_data = to_fp16(coord_data, name + 'coord_data_fp16')
        name = name + '_'
        rel = self.relative_coord(self.sample_coord(name, coord_data, kernel_size), coord_data, coord_channels, kernel_size)
        weights = self.mlp(rel, name, in_channels=coord_channels, channel_list=channel_list, norm=norm, no_bias=False)
        taps = kernel_size * kernel_size
        neighbours = mx.reshape(self.sample_data(name, data, kernel_size),
                                shape=(self.num_batch, data_channels, taps, self.H, self.W))
        return mx.reshape(neighbours * weights, shape=(self.num_batch, -1, self.H, self.W))
