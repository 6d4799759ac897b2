#!/bin/bash
# round 6, third session: does the runtime's hardware-queue limit (GPU_MAX_HW_QUEUES, default 4) bind the batches in flight?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/r6o; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "frames/s", d["ms_per_step"], d["config"]["results_sha256_all_steps"])'
for rep in 1 2; do
for q in "" 8 2; do
  for n in 3 4; do
    echo -n "GPU_MAX_HW_QUEUES=${q:-default} --inflight $n: "
    if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
    timeout -s KILL 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --backbone-reps 0 --min-timed-s 5 --inflight $n 2>/dev/null | tail -1 | python -c "$P"
  done
done
done
