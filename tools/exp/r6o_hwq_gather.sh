#!/bin/bash
# round 6, third session: the multi-GPU code path (RD_BENCH_GATHER=1: RCCL all_gather on one communication stream per pipeline) on ONE GPU --
# do its extra streams share hardware queues with the launch streams (GPU_MAX_HW_QUEUES, default 4)?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/r6o; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "frames/s", d["ms_per_step"], d["config"]["results_sha256_all_steps"], d["config"]["rccl_version"])'
for rep in 1 2 3; do
for g in "" 1; do
for q in "" 8; do
    echo -n "gather=${g:-0} GPU_MAX_HW_QUEUES=${q:-default}: "
    if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
    if [ -n "$g" ]; then export RD_BENCH_GATHER=1; else unset RD_BENCH_GATHER; fi
    MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 timeout -s KILL 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --backbone-reps 0 --min-timed-s 5 2>/dev/null | tail -1 | python -c "$P"
done
done
done
