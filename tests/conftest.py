import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size case")


class Backend:
    """Where a test runs the product's HIP sources: 'emu' = hipemu CPU build (test infrastructure, CPU tier),
    'hip' = the real librangedet_hip.so on cuda:0 (GPU tier)."""

    def __init__(self, name):
        self.name = name
        self._keep = []
        if name == "emu":
            from emu_util import emu_lib, NumpyAllocator
            self.lib, self.alloc = emu_lib(), NumpyAllocator()
        else:
            import torch
            assert torch.cuda.is_available(), "gpu-marked test without a GPU"
            from rangedet_amd import lib as rdlib
            from rangedet_amd.runtime import TorchAllocator
            self.lib, self.alloc = rdlib.get_lib(), TorchAllocator("cuda:0")

    def up(self, arr):
        buf = self.alloc.upload(np.ascontiguousarray(arr))
        self._keep.append(buf)  # tests pass be.ptr(be.up(x)) inline: keep the buffer alive until the test ends
        return buf

    def empty(self, nbytes):
        return self.alloc.alloc(nbytes, zero=True)

    def ptr(self, buf):
        return None if buf is None else self.alloc.ptr(buf)

    def down(self, buf, dtype, shape):
        self.alloc.sync()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if self.name == "emu":
            return buf[:n].view(dtype).reshape(shape).copy()
        return buf[:n].cpu().numpy().view(dtype).reshape(shape).copy()

    @property
    def stream(self):
        return self.alloc.stream


_BACKENDS = {}


@pytest.fixture
def be(request):
    name = request.param
    if name not in _BACKENDS:
        _BACKENDS[name] = Backend(name)
    _BACKENDS[name]._keep.clear()
    return _BACKENDS[name]


BOTH = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
HIP_ONLY = [pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
