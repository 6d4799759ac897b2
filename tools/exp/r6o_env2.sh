#!/bin/bash
# round 6, third session: more runtime knobs of the dispatch path (names from the strings of libamdhip64.so), one at a time against the default
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "frames/s", d["ms_per_step"], d["config"]["results_sha256_all_steps"])'
for rep in 1 2; do
for kv in "" "ROC_SYSTEM_SCOPE_SIGNAL=0" "AMD_OPT_FLUSH=0" "DEBUG_HIP_DYNAMIC_QUEUES=1" "GPU_STREAMOPS_CP_WAIT=1" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "DEBUG_CLR_MAX_BATCH_SIZE=1" "ROC_USE_FGS_KERNARG=0" "ROC_AQL_QUEUE_SIZE=65536"; do
    echo -n "${kv:-default}: "
    env $kv timeout -s KILL 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --backbone-reps 0 --min-timed-s 5 2>/dev/null | tail -1 | python -c "$P" || echo failed
done
done
