#!/bin/bash
# round 6, third session: what of the multi-GPU code path (RD_BENCH_GATHER=1 on ONE GPU) costs frames/s?
#   off = no communicator; init = communicator up, no per-step collective; pack = pack copies + event + communication-stream wait, no collective;
#   copy = the collective replaced by a device copy on the communication stream; full = the RCCL all_gather
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/r6o; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "frames/s", d["ms_per_step"], d["config"]["results_sha256_all_steps"])'
for rep in 1 2; do
for m in ${MODES:-off init pack copy full}; do
    echo -n "mode $m: "
    unset RD_BENCH_GATHER RD_BENCH_GATHER_MODE
    [ $m != off ] && export RD_BENCH_GATHER=1
    [ $m != off ] && [ $m != full ] && export RD_BENCH_GATHER_MODE=$m
    MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 timeout -s KILL 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --backbone-reps 0 --min-timed-s 5 $EXTRA 2>$O/gc_err.txt | tail -1 | python -c "$P" || tail -3 $O/gc_err.txt
done
done
