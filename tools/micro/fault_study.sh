#!/bin/bash
# The packed-fp32 fault of round 5 in one call (run THROUGH gpurun; writes gpurun_out/fault/study.txt, copied to profiles/ by hand):
#   1. which packed form, next to which load          (pkform_test.py)
#   2. which instruction class of the co-resident wave (aggr_test.py)
#   3. the library's NMS chain next to its own forward, this build and -- if tools/micro/librangedet_hip_slp.so exists -- a build WITH
#      the SLP vectoriser (RD_ALLOW_PACKED_SWAP=1 RD_EXTRA_HIPCC_FLAGS=-fslp-vectorize python -m rangedet_amd.build --force)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/fault; mkdir -p $O
{
  echo "== rocm-smi / device"; python -c "import torch;print(torch.cuda.get_device_name(0), torch.version.hip)"
  echo; echo "== 1. packed forms x loads (tools/micro/pkform_test.py 6)"; timeout 300 python tools/micro/pkform_test.py 6 2>&1 | grep -v amdgpu.ids
  echo; echo "== 2. instruction class of the co-resident wave (tools/micro/aggr_test.py 5)"; timeout 300 python tools/micro/aggr_test.py 5 2>&1 | grep -v amdgpu.ids
  echo; echo "== 3. library: batched WNMS chain replayed next to another batch's forward (tools/nms_race.py 40)"
  echo "-- this build:"; timeout 300 python tools/nms_race.py 40 2>&1 | grep "differ from\|idle:" | cut -c1-200
  if [ -f tools/micro/librangedet_hip_slp.so ]; then
    echo "-- the same sources built with the SLP vectoriser:"; RANGEDET_HIP_LIB=$PWD/tools/micro/librangedet_hip_slp.so timeout 300 python tools/nms_race.py 40 2>&1 | grep "differ from\|stage\|idle:" | cut -c1-330 | head -12
    echo "-- per plan step as the only concurrent load (LOAD=sweep), SLP build:"; RANGEDET_HIP_LIB=$PWD/tools/micro/librangedet_hip_slp.so LOAD=sweep LREP=12 TIE=stable timeout 300 python tools/nms_race.py 6 2>&1 | grep "^step" | python -c "
import sys, re, collections
z, nz = collections.Counter(), collections.Counter()
for l in sys.stdin:
    m = re.match(r'step +\\d+ (\\S+).*cout (\\S+) W.*: (\\d+) of', l)
    k = '%-10s cout %s' % (m.group(1), m.group(2))
    (z if m.group(3) == '0' else nz)[k] += 1
for k in sorted(set(z) | set(nz)): print('    %-22s differing in %2d step(s), clean in %2d' % (k, nz[k], z[k]))
"
  fi
} > $O/study.txt 2>&1
tail -5 $O/study.txt
