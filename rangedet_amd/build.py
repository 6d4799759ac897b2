"""Build librangedet_hip.so (gfx950) in-tree with hipcc.  ``python -m rangedet_amd.build [--force] [--dev]``.

The default build is the RELEASE library: no development switch, no getenv (csrc/rd_common.h).  ``--dev`` builds
librangedet_hip_dev.so with -DRD_DEV_SWITCHES next to it (the A/B library of tools/exp/ab.sh: load it with RANGEDET_HIP_LIB=...)."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "rd_api.hip")
OUT = os.path.join(HERE, "librangedet_hip.so")
# -fno-slp-vectorize (device side): the SLP vectoriser turns the cross products of the rotated-box code into packed-fp32 instructions
# whose second source has its halves SWAPPED (v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]), and on the MI355X boxes of this project that
# form returns a wrong low half in lanes 48-63 whenever ANOTHER wave of the same SIMD is issuing MFMA instructions -- i.e. whenever the
# weighted NMS of one batch overlaps the convolutions of the next (reproducer: tools/micro/pkform_test.py, aggr_test.py).
# packed_swap_lint() below fails the build if such an instruction is left in the code object.
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result"]
CODEGEN_FLAGS = ["-Xarch_device", "-fno-slp-vectorize"]     # (part of source_hash(): they change the device code)
FLAGS = BASE_FLAGS + CODEGEN_FLAGS
OUT_DEV = os.path.join(HERE, "librangedet_hip_dev.so")


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _objdump():
    """llvm-objdump of the SAME ROCm install as the compiler (RD_OBJDUMP overrides; ADVICE r5: the path was hard-coded)."""
    if os.environ.get("RD_OBJDUMP"):
        return os.environ["RD_OBJDUMP"]
    roots = [os.environ.get("ROCM_PATH"), os.path.dirname(os.path.dirname(os.path.realpath(_hipcc()))), "/opt/rocm"]
    for r in roots:
        if r and os.path.exists(os.path.join(r, "lib", "llvm", "bin", "llvm-objdump")):
            return os.path.join(r, "lib", "llvm", "bin", "llvm-objdump")
    found = shutil.which("llvm-objdump")
    if found:
        return found
    raise RuntimeError("packed_swap_lint: no llvm-objdump next to %s (set ROCM_PATH or RD_OBJDUMP)" % _hipcc())


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources: what a measured profile was taken on (profiles/*pmc_traffic.json
    carry it, bench.py only quotes a measured HBM traffic figure whose hash matches the sources it runs)."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(CODEGEN_FLAGS).encode())           # code-generation flags added since the first measured profile
    for f in sorted(glob.glob(os.path.join(HERE, "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def stale():
    return _stale(OUT)


_PK = re.compile(r"\b(v_pk_(?:mul|add|fma)_f32)\b(.*)")
_SEL = re.compile(r"op_sel:\[([01,]+)\]")
_SELHI = re.compile(r"op_sel_hi:\[([01,]+)\]")


def swapped_sources(line):
    """Source operands (0-based) of a packed-fp32 instruction whose halves are swapped: low result takes the high register, high result
    the low one.  None if the line is no packed-fp32 arithmetic instruction."""
    m = _PK.search(line)
    if not m:
        return None
    n = 3 if m.group(1) == "v_pk_fma_f32" else 2
    lo, hi = _SEL.search(m.group(2)), _SELHI.search(m.group(2))
    lo = [int(v) for v in lo.group(1).split(",")] if lo else [0] * n       # defaults: low half <- low register, high <- high
    hi = [int(v) for v in hi.group(1).split(",")] if hi else [1] * n
    return [i for i in range(n) if lo[i] == 1 and hi[i] == 0]


def packed_swap_lint(so=OUT):
    """Disassemble the gfx950 code object inside `so` and return [(kernel, instruction)] of every packed-fp32 instruction with a swapped
    SECOND or THIRD source (measured wrong next to MFMA waves: src1; src2 was not measured and is refused as well; src0 swapped measured
    clean).  Raises if the tools are missing -- a library that cannot be checked is not shipped."""
    td = tempfile.mkdtemp(prefix="rd_lint_")
    try:
        tmp = os.path.join(td, "lib.so")
        shutil.copy(so, tmp)
        OBJDUMP = _objdump()
        subprocess.check_call([OBJDUMP, "--offloading", tmp], cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = [f for f in glob.glob(os.path.join(td, "lib.so.*")) if "gfx950" in f]
        if len(objs) != 1:
            raise RuntimeError("packed_swap_lint: expected one gfx950 code object in %s, found %r" % (so, objs))
        dis = subprocess.Popen([OBJDUMP, "-d", "--mcpu=gfx950", objs[0]], stdout=subprocess.PIPE, text=True)
        found, kernel = [], "?"
        for line in dis.stdout:
            if line.endswith(">:\n"):
                kernel = line.split("<", 1)[1][:-3]
            elif "v_pk_" in line:
                sw = swapped_sources(line)
                if sw and any(i >= 1 for i in sw):
                    found.append((kernel, line.split("//")[0].strip()))
        if dis.wait() != 0:
            raise RuntimeError("packed_swap_lint: llvm-objdump failed on %s" % objs[0])
        return found
    finally:
        shutil.rmtree(td, ignore_errors=True)


def _stale(out):
    if not os.path.exists(out):
        return True
    deps = glob.glob(os.path.join(HERE, "csrc", "*")) + [os.path.join(HERE, "..", "include", "rangedet_hip.h")]
    return os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps)


def build(force=False, verbose=True, dev=False):
    """dev=True: librangedet_hip_dev.so with -DRD_DEV_SWITCHES (the environment-driven A/B switches of csrc/rd_common.h)."""
    out = OUT_DEV if dev else OUT
    if not force and not _stale(out):
        return out
    # (a per-process temporary name: several ranks that find the library stale at once must not clobber each other's output)
    tmp_out = "%s.new.%d" % (out, os.getpid())
    cmd = [_hipcc()] + FLAGS + (["-DRD_DEV_SWITCHES"] if dev else []) + os.environ.get("RD_EXTRA_HIPCC_FLAGS", "").split() + [SRC, "-o", tmp_out]   # e.g. -DRD_CONV3_DEV
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        subprocess.check_call(cmd)
        bad = packed_swap_lint(tmp_out)
        if bad and not os.environ.get("RD_ALLOW_PACKED_SWAP"):        # (the switch exists for the A/B of the fault itself)
            raise RuntimeError("%s: %d packed-fp32 instructions with swapped source halves (wrong next to MFMA waves on gfx950, "
                               "DESIGN.md 6.4), first: %s in %s" % (os.path.basename(out), len(bad), bad[0][1], bad[0][0]))
        os.replace(tmp_out, out)
    finally:
        if os.path.exists(tmp_out):
            os.remove(tmp_out)
    return out


if __name__ == "__main__":
    if "--lint" in sys.argv:
        hits = packed_swap_lint()
        for k, ins in hits[:20]:
            print(k[:80], "|", ins)
        print("%d packed-fp32 instructions with a swapped second/third source" % len(hits))
        sys.exit(1 if hits else 0)
    build(force="--force" in sys.argv, dev="--dev" in sys.argv)
