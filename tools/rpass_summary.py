"""Per-kernel register / spill summary of a hipcc build log made with -Rpass-analysis=kernel-resource-usage
(RD_EXTRA_HIPCC_FLAGS="-Rpass-analysis=kernel-resource-usage" python -m rangedet_amd.build --force > log 2>&1).
usage: python tools/rpass_summary.py LOG [substring of the demangled kernel name]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: Function Name: ", txt)[1:]
names = [b.split()[0] for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")


def g(b, k):
    m = re.search(k + r": (\d+)", b)
    return int(m.group(1)) if m else -1


for b, d in zip(blocks, dem):
    if want not in d:
        continue
    d = re.sub(r"^void rd::", "", d)
    d = re.sub(r"\(rd::\w+\)$", "", d)
    print("%-90s VGPR %3d AGPR %3d spillV %3d spillS %3d scratch %4d LDS %6d occ %d" % (
        d[:90], g(b, "VGPRs"), g(b, "AGPRs"), g(b, "VGPRs Spill"), g(b, "SGPRs Spill"), g(b, r"ScratchSize \[bytes/lane\]"),
        g(b, r"LDS Size \[bytes/block\]"), g(b, r"Occupancy \[waves/SIMD\]")))
