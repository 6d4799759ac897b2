import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _auto_parallel(config):
    """The CPU tier (`-m "not gpu"`: the unmodified HIP sources under the CPU emulator) takes 22 min in one process and 6.5 min in four.
    When exactly that tier is asked for without any distribution option and pytest-xdist is importable, run it on four workers -- the same
    as `-n 4` on the command line (RD_PYTEST_SERIAL=1 keeps one process).  The GPU tier is never touched: one process, one GPU, and the
    library the tests load is mapped into THAT process."""
    opt = config.option
    if os.environ.get("RD_PYTEST_SERIAL") or hasattr(config, "workerinput") or not config.pluginmanager.hasplugin("xdist"):
        return
    if getattr(opt, "numprocesses", None) is not None or getattr(opt, "dist", "no") != "no" or getattr(opt, "tx", None):
        return
    if (getattr(opt, "markexpr", "") or "").strip() != "not gpu" or getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False):
        return
    n = min(4, os.cpu_count() or 1)
    if n < 2:
        return
    try:                                    # build the emulator library ONCE, here, before the workers start
        from emu_util import emu_lib
        emu_lib()
    except Exception:                       # (a broken build fails in the tests that need it, with their own message)
        pass
    opt.numprocesses, opt.dist, opt.tx = n, "load", ["popen"] * n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size case")
    _auto_parallel(config)


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
    """'emu' / 'hip' as above; 'emu_late' = the emu backend with LDS-DMA transfers landing as LATE as the hardware allows (only when a
    counted s_waitcnt of the issuing wave forces them, tests/emu/hip/hip_runtime.h dma_late): a wait that is too weak reads stale LDS
    deterministically.  (The default emu mode lands them when issued -- the earliest the hardware allows.)"""
    name = request.param
    late = name == "emu_late"
    key = "emu" if late else name
    if key not in _BACKENDS:
        _BACKENDS[key] = Backend(key)
    b = _BACKENDS[key]
    b._keep.clear()
    if not late:
        yield b
        return
    b.lib.cdll.hipemu_set_dma_late(1)
    try:
        yield b
    finally:
        b.lib.cdll.hipemu_set_dma_late(0)


BOTH = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
HIP_ONLY = [pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
# the kernels that move data with LDS-DMA under counted waits (persistent 3x3 conv, fused block, streaming 1x1 conv, generic tap conv, Meta-Kernel)
# also run with the transfers landing late
WITH_LATE_DMA = BOTH + [pytest.param("emu_late", id="emu-late-dma")]
