"""Test-time input transform chain on the device: the build's counterpart of the reference's loader-side transforms
(rangedet/core/input.py: LoadRecord, ProcessMissValue, SepAndClipData, GetUnnormalizedRange, NormData, GetCoordinates,
CombineData, PadData, TransposeData, GenerateFPNTarget, TransAndReshape; constants config:245-282,71).

Only the raw record arrays (range_image (H,W,4), pc_vehicle_frame (H,W,3), inclination (H,)) go to the GPU; one fused
kernel (csrc/k_input.h, rd_input_transform) writes the named float32 tensors the graph consumes.
"""
import ctypes

import numpy as np

from . import lib as rdlib
from . import synth

ORDER = ['range_value', 'intensity', 'elongation', 'pc_vehicle_frame_x', 'pc_vehicle_frame_y', 'pc_vehicle_frame_z',
         'inclination', 'azimuth']


class InputNorm(ctypes.Structure):   # rd_input_norm_t
    _fields_ = [("clip_lo", ctypes.c_float * 7), ("clip_hi", ctypes.c_float * 7), ("mean", ctypes.c_float * 8),
                ("sd", ctypes.c_float * 8), ("interval_lo", ctypes.c_float * 3), ("interval_hi", ctypes.c_float * 3)]


def make_norm(clip=None, norm=None, interval=None, strides=(1, 2, 4)):
    clip, norm, interval = clip or synth.CLIP, norm or synth.NORM, interval or synth.INTERVAL
    n = InputNorm()
    for i, name in enumerate(ORDER):
        if i < 7:
            n.clip_lo[i], n.clip_hi[i] = clip[name]
        n.mean[i] = norm[name][0]
        n.sd[i] = np.float32(norm[name][1] ** 0.5)
    for l, s in enumerate(strides):
        n.interval_lo[l], n.interval_hi[l] = interval[s]
    return n


class DeviceInputTransform:
    def __init__(self, pad_hw=(64, 2656), lib=None, alloc=None, clip=None, norm=None, interval=None):
        from .runtime import TorchAllocator
        self.L = lib or rdlib.get_lib()
        self.A = alloc or TorchAllocator()
        self.pad_hw = tuple(pad_hw)
        self.norm = make_norm(clip, norm, interval)

    def __call__(self, records):
        """records: list of dicts with 'range_image', 'pc_vehicle_frame', 'inclination' (numpy) -> dict of device tensors."""
        A, L = self.A, self.L
        B = len(records)
        H, W, _ = records[0]['range_image'].shape
        Hp, Wp = self.pad_hw
        ri = A.upload(np.stack([np.asarray(r['range_image'], np.float32) for r in records]))
        pc = A.upload(np.stack([np.asarray(r['pc_vehicle_frame'], np.float32) for r in records]))
        inc = A.upload(np.stack([np.asarray(r['inclination'], np.float32) for r in records]))
        npx = Hp * Wp
        shapes = {'input_data': (B, 8, Hp, Wp), 'coord_s1': (B, 3, Hp, Wp)}
        for s in (1, 2, 4):
            shapes['pc_vehicle_frame_s%d' % s] = (B, npx // s, 3)
            shapes['range_image_mask_s%d' % s] = (B, npx // s)
        bufs = {k: A.alloc(int(np.prod(v)) * 4) for k, v in shapes.items()}
        st = A.stream_ptr(None) if hasattr(A, "stream_ptr") else A.stream
        L.call("rd_input_transform", A.ptr(ri), A.ptr(pc), A.ptr(inc), ctypes.addressof(self.norm), B, H, W, Hp, Wp,
               A.ptr(bufs['input_data']), A.ptr(bufs['coord_s1']), A.ptr(bufs['pc_vehicle_frame_s1']),
               A.ptr(bufs['pc_vehicle_frame_s2']), A.ptr(bufs['pc_vehicle_frame_s4']), A.ptr(bufs['range_image_mask_s1']),
               A.ptr(bufs['range_image_mask_s2']), A.ptr(bufs['range_image_mask_s4']), st)
        return {k: A.view_f32(bufs[k], shapes[k]) for k in shapes}
