"""Test-only helpers: bind the hipemu (CPU) build of the product's HIP sources and move numpy data in and out of
the library's channels-last layouts.  TEST INFRASTRUCTURE ONLY -- never imported by rangedet_amd/."""
import os
import subprocess

import numpy as np

from rangedet_amd import lib as rdlib

_HERE = os.path.dirname(os.path.abspath(__file__))
EMU_SO = os.path.join(_HERE, "emu", "librangedet_emu.so")
_EMU = None


def emu_lib():
    global _EMU
    if _EMU is None:
        srcs = [os.path.join(_HERE, "..", "rangedet_amd", "csrc", f) for f in os.listdir(os.path.join(_HERE, "..", "rangedet_amd", "csrc"))]
        srcs.append(os.path.join(_HERE, "emu", "hip", "hip_runtime.h"))
        newest = max(os.path.getmtime(s) for s in srcs)
        if not os.path.exists(EMU_SO) or os.path.getmtime(EMU_SO) < newest:
            import fcntl
            with open(EMU_SO + ".lock", "w") as lk:           # pytest-xdist workers: one of them builds, the others wait and re-check
                fcntl.flock(lk, fcntl.LOCK_EX)
                if not os.path.exists(EMU_SO) or os.path.getmtime(EMU_SO) < newest:
                    subprocess.check_call([os.path.join(_HERE, "emu", "build_emu.sh")])
        _EMU = rdlib.Lib(EMU_SO)
    return _EMU


def f32_to_bf16_bits(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = u + 0x7FFF + ((u >> 16) & 1)
    return (r >> 16).astype(np.uint16)


def bf16_bits_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def bf16_round(a):
    return bf16_bits_to_f32(f32_to_bf16_bits(a))


def f32_to_f16_bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).astype(np.float16).view(np.uint16)


def f16_bits_to_f32(b):
    return np.ascontiguousarray(b).view(np.float16).astype(np.float32)


def h16_round(a, dtype):
    """Round float32 values to the library's 16-bit element type `dtype` (RD_BF16 / RD_F16), back as float32; RD_F32 unchanged."""
    if dtype == rdlib.RD_BF16:
        return bf16_round(a)
    if dtype == rdlib.RD_F16:
        return f16_bits_to_f32(f32_to_f16_bits(a))
    return np.asarray(a, dtype=np.float32)


def to_nhwc(x_nchw, dtype, cstride=None, coff=0):
    """(B,C,H,W) float32 -> channels-last buffer (B,H,W,cstride) of the library dtype (zeros elsewhere)."""
    B, C, H, W = x_nchw.shape
    cs = cstride or C
    buf = np.zeros((B, H, W, cs), dtype=np.float32)
    buf[..., coff:coff + C] = np.transpose(x_nchw, (0, 2, 3, 1))
    return f32_to_bf16_bits(buf) if dtype == rdlib.RD_BF16 else f32_to_f16_bits(buf) if dtype == rdlib.RD_F16 else buf


def from_nhwc(buf, dtype, C, coff=0):
    a = bf16_bits_to_f32(buf) if dtype == rdlib.RD_BF16 else f16_bits_to_f32(buf) if dtype == rdlib.RD_F16 else buf
    return np.transpose(a[..., coff:coff + C], (0, 3, 1, 2)).copy()


def empty_nhwc(B, H, W, cs, dtype):
    return np.zeros((B, H, W, cs), dtype=np.uint16 if dtype in rdlib.H16 else np.float32)


class NumpyAllocator:
    """Host-memory stand-in for rangedet_amd.runtime.TorchAllocator, for the hipemu library only (tests)."""
    stream = None

    def alloc(self, nbytes, zero=False):
        return np.zeros(max(int(nbytes), 16) + 64, dtype=np.uint8)

    def upload(self, arr):
        return np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy()

    def ptr(self, buf):
        return buf.ctypes.data

    def view_f32(self, buf, shape):
        n = int(np.prod(shape))
        return buf[: n * 4].view(np.float32).reshape(shape)

    def view_i32(self, buf, shape):
        n = int(np.prod(shape))
        return buf[: n * 4].view(np.int32).reshape(shape)

    def to_numpy(self, t):
        return np.array(t)

    def as_device_f32(self, x):
        return np.ascontiguousarray(np.asarray(x), dtype=np.float32)

    def assign(self, dst_view, src):
        dst_view[...] = np.asarray(src).reshape(dst_view.shape)

    def sync(self):
        pass


def small_shapes(H, W):
    shapes = {'input_data': (8, H, W), 'coord_s1': (3, H, W)}
    for s in (1, 2, 4):
        shapes['pc_vehicle_frame_s%d' % s] = (H * W // s, 3)
        shapes['range_image_mask_s%d' % s] = (H * W // s,)
    return shapes
