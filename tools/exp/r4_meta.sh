#!/bin/bash
# round 4: experimental forms of meta16_kernel (k_meta.h template parameter V) -- tools/micro/meta_v_bench.py with a list of forms
#   r4_meta.sh OUT "X(0) X(11) ..." [extra hipcc flags]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p "gpurun_out/$1"
MV_FLAGS="-DMV_LIST=\"$2\" $3" timeout 500 python tools/micro/meta_v_bench.py 8 30 2>&1 | grep -v amdgpu.ids | tee "gpurun_out/$1/meta_v.txt" | grep -v "^round"
