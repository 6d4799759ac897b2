"""Build librangedet_hip.so (gfx950) in-tree with hipcc.  ``python -m rangedet_amd.build [--force]``."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "rd_api.hip")
OUT = os.path.join(HERE, "librangedet_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result"]


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources: what a measured profile was taken on (profiles/*pmc_traffic.json
    carry it, bench.py only quotes a measured HBM traffic figure whose hash matches the sources it runs)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(HERE, "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def stale():
    if not os.path.exists(OUT):
        return True
    deps = glob.glob(os.path.join(HERE, "csrc", "*")) + [os.path.join(HERE, "..", "include", "rangedet_hip.h")]
    return os.path.getmtime(OUT) < max(os.path.getmtime(d) for d in deps)


def build(force=False, verbose=True):
    if not force and not stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + os.environ.get("RD_EXTRA_HIPCC_FLAGS", "").split() + [SRC, "-o", OUT]   # e.g. -DRD_CONV3_DEV
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
