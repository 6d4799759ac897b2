#!/bin/bash
# the whole GPU tier + smoke, then the round's profile (tag in $1)
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/exp/gpu_tests.sh
bash tools/profile_round.sh "$1" > gpurun_out/p_round.log 2>&1
tail -c 600 gpurun_out/p/bench.json
