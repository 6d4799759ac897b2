#!/bin/bash
# round 6, third session: more batches in flight with enough hardware queues (GPU_MAX_HW_QUEUES=8)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "frames/s", d["ms_per_step"], d["config"]["results_sha256_all_steps"])'
for rep in 1 2; do
for n in 3 4 5 6; do
    echo -n "GPU_MAX_HW_QUEUES=8 --inflight $n: "
    GPU_MAX_HW_QUEUES=8 timeout -s KILL 200 python bench.py --steps 100 --warmup 6 --no-cpu-baseline --backbone-reps 0 --min-timed-s 5 --inflight $n 2>/dev/null | tail -1 | python -c "$P"
done
echo -n "default queues --inflight 3: "; timeout -s KILL 200 python bench.py --steps 100 --warmup 6 --no-cpu-baseline --backbone-reps 0 --min-timed-s 5 --inflight 3 2>/dev/null | tail -1 | python -c "$P"
done
