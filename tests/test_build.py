"""The build's own checks (rangedet_amd/build.py): the code object must not contain the packed-fp32 form that is wrong next to MFMA waves on
gfx950.  No GPU needed: the in-tree library is disassembled."""
import os

import pytest

from rangedet_amd import build as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_swapped_sources_parser():
    f = B.swapped_sources
    assert f("\tv_pk_mul_f32 v[18:19], v[20:21], v[18:19] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]// 0000757B8: D3B15212") == [1]
    assert f("\tv_pk_add_f32 v[2:3], v[8:9], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]") == [1]
    assert f("\tv_pk_fma_f32 v[28:29], v[20:21], v[18:19], v[22:23] op_sel:[0,1,0] op_sel_hi:[1,0,1]") == [1]
    assert f("\tv_pk_fma_f32 v[28:29], v[20:21], v[18:19], v[22:23] op_sel:[0,0,1] op_sel_hi:[1,1,0]") == [2]
    assert f("\tv_pk_mul_f32 v[28:29], v[20:21], v[18:19] op_sel:[1,0] op_sel_hi:[0,1]") == [0]          # first source swapped: measured clean
    assert f("\tv_pk_mul_f32 v[22:23], v[22:23], v[18:19]") == []
    assert f("\tv_pk_mul_f32 v[24:25], v[24:25], s[0:1] op_sel_hi:[1,0]") == []                           # low half broadcast, not a swap
    assert f("\tv_pk_mul_f32 v[24:25], v[24:25], v[0:1] op_sel:[0,1]") == []                              # high half broadcast (op_sel_hi defaults to 1)
    assert f("\tv_pk_add_f32 v[18:19], v[18:19], v[20:21] neg_lo:[0,1] neg_hi:[0,1]") == []
    assert f("\tv_pk_mov_b32 v[22:23], v[20:21], v[24:25] op_sel:[1,0]") is None                          # not arithmetic (measured clean)
    assert f("\tv_pk_max_i16 v1, v2, v3") is None and f("\tv_mul_f32_e32 v1, v2, v3") is None


def test_library_has_no_swapped_packed_fp32():
    """The shipped library is checked, not just the flags it was built with (build() refuses to install a library that fails this)."""
    if not os.path.exists(B.OUT):
        B.build()
    B._objdump()          # (raises with a clear message when no llvm-objdump belongs to the compiler's ROCm install)
    assert B.packed_swap_lint() == []
    assert "-fno-slp-vectorize" in B.FLAGS


@pytest.mark.gpu
def test_packed_fp32_fault_reproducer_on_this_gpu(tmp_path):
    """the hardware claim behind -fno-slp-vectorize as a record of the GPU tier.  Builds the
    stand-alone reproducer (tools/micro/victim.hip -DFAULT_REPRO_MAIN: two streams, an MFMA spin kernel next to single-instruction
    victims, nothing of the library) on the box and runs it.  Hard requirements: the DEFAULT packed form never differs, and on the idle
    GPU nothing differs (the victim kernels are right).  The fault itself -- wrong low halves of `v_pk_mul_f32 ... op_sel:[0,1]
    op_sel_hi:[1,0]` in lanes 48-63 next to MFMA waves -- is REPORTED: reproduced = pass, not reproduced on this box = xfail with the
    counts, so that either outcome is in the driver's log."""
    import re
    import subprocess
    exe = str(tmp_path / "fault_repro")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-DFAULT_REPRO_MAIN", os.path.join(ROOT, "tools", "micro", "victim.hip"), "-o", exe],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    print(r.stdout)
    rows = {}
    for line in r.stdout.splitlines():
        m = re.search(r"^(next to \S+ waves|idle GPU)\s+.*op_sel_hi:\[1,0\]: lo \[(\d+), (\d+), (\d+), (\d+)\] hi \[(\d+), (\d+), (\d+), (\d+)\]\s+default form: lo \[(\d+), (\d+), (\d+), (\d+)\] hi \[(\d+), (\d+), (\d+), (\d+)\]", line)
        if m:
            v = [int(x) for x in m.groups()[1:]]
            rows["idle" if m.group(1).startswith("idle") else "mfma"] = (v[:8], v[8:])
    assert set(rows) == {"idle", "mfma"}, r.stdout
    assert sum(rows["idle"][0]) == 0 and sum(rows["idle"][1]) == 0, ("the victim kernel differs on an idle GPU", rows)
    assert sum(rows["mfma"][1]) == 0, ("the DEFAULT packed form differs next to MFMA waves: the lint's instruction list is incomplete", rows)
    swapped = rows["mfma"][0]
    if sum(swapped) == 0:
        pytest.xfail("packed-fp32 fault NOT reproduced on this box (0 wrong halves next to MFMA waves): %r" % (rows,))
    assert swapped[3] > 0 and sum(swapped[4:]) == 0, ("expected wrong LOW halves in lanes 48-63 only", rows)


def test_lint_covers_every_packed_fp32_opcode_of_the_library():
    """Lint completeness: the refused forms are v_pk_{mul,add,fma}_f32 with a swapped second / third source.  Every packed-fp32 ARITHMETIC
    opcode that occurs in the shipped code object must be one the lint parses (a new opcode -- e.g. a future v_pk_sub_f32 -- would
    otherwise pass unchecked)."""
    import glob
    import shutil
    import subprocess
    import tempfile
    td = tempfile.mkdtemp(prefix="rd_lintcov_")
    try:
        tmp = os.path.join(td, "lib.so")
        shutil.copy(B.OUT, tmp)
        subprocess.check_call([B._objdump(), "--offloading", tmp], cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        obj = [f for f in glob.glob(os.path.join(td, "lib.so.*")) if "gfx950" in f][0]
        dis = subprocess.run([B._objdump(), "-d", "--mcpu=gfx950", obj], capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(td, ignore_errors=True)
    import re
    ops = set(re.findall(r"\b(v_pk_[a-z0-9_]+_f32)\b", dis))
    arith = {o for o in ops if o != "v_pk_mov_b32"}
    assert arith <= {"v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32"}, "packed-fp32 opcodes the lint does not parse: %r" % sorted(arith - {"v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32"})
    assert all(B.swapped_sources("\t" + l.split("//")[0].strip()) is not None for l in dis.splitlines() if re.search(r"\bv_pk_(mul|add|fma)_f32\b", l))
