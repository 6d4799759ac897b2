"""The Winograd F(2,3) 3x3 conv EXPERIMENT of round 6 (tools/micro/k_wino.h; measured slower than the direct kernel and not part of the
product, profiles/EXPERIMENTS.md): its parity run on the CPU emulator, so that the numbers in the log belong to a kernel that computes
the right thing -- early-DMA and late-DMA models of the LDS-DMA transfers (counted s_waitcnt vmcnt(N) waits)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("late", ["0", "1"])
def test_winograd_experiment_kernel_parity_on_the_emulator(late):
    env = dict(os.environ, HIPEMU_DMA_LATE=late, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "micro", "wino_dev.py"), "emu"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and ("emu OK (dma late %s)" % late) in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
