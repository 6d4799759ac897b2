#!/bin/bash
# round 6, third session: what does the multi-GPU code path (RD_BENCH_GATHER=1 on ONE GPU) cost in frames/s?
#   off = no communicator; eager = the collective enqueued with the batch, its communication stream waiting for the pack's event (first session);
#   full = the shipping form: the host issues the collective once the batch's event has fired.
# (The bisect that found the cost -- modes init / pack / copy / packonly / evonly / evwait / evlate -- ran at commit 4803186,
#  profiles/r06o_gather_path_cost_bisect.txt.)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/r6o; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "frames/s", d["ms_per_step"], d["config"]["results_sha256_all_steps"])'
for rep in 1 2; do
for m in ${MODES:-off eager full}; do
    echo -n "mode $m: "
    unset RD_BENCH_GATHER RD_BENCH_GATHER_MODE
    [ $m != off ] && export RD_BENCH_GATHER=1
    [ $m = eager ] && export RD_BENCH_GATHER_MODE=eager
    MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 timeout -s KILL 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --backbone-reps 0 --min-timed-s 5 $EXTRA 2>$O/gc_err.txt | tail -1 | python -c "$P" || tail -3 $O/gc_err.txt
done
done
