#!/bin/bash
# round 6, power probe: what the matrix cores sustain under the power cap by operand statistics / roles / shape / source, and the
# production 128->128 and 64->64 convs under the same sampler.   gpurun -- 'bash tools/exp/r6g.sh [long]'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/r6g; mkdir -p $O
timeout -s KILL 600 tools/micro/mfma_power.bin $1 2>&1 | tee $O/mfma_power_$1.txt
if [ -z "$1" ]; then
timeout -s KILL 600 python tools/conv_clock.py 2656 128 128 8 2>&1 | grep -v amdgpu.ids | tee $O/conv_clock.txt
timeout -s KILL 600 python tools/conv_clock.py 2656 64 64 8 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_clock.txt
fi
