#!/bin/bash
# round 6, third session: runtime knobs of the dispatch path (one variable at a time, alternating with the default)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/r6o; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "frames/s", d["ms_per_step"], d["config"]["results_sha256_all_steps"])'
for rep in 1 2; do
for kv in "" "HIP_FORCE_DEV_KERNARG=1" "HSA_ENABLE_INTERRUPT=0" "HIP_FORCE_DEV_KERNARG=0" "GPU_MAX_HW_QUEUES=3" "AMD_SERIALIZE_KERNEL=0 HSA_ENABLE_SDMA=0"; do
    echo -n "${kv:-default}: "
    env $kv timeout -s KILL 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --backbone-reps 0 --min-timed-s 6 2>/dev/null | tail -1 | python -c "$P"
done
done
