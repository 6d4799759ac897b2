#!/bin/bash
# round 6, first GPU call of the release build: hipGraph replay (digest equality, enqueue cost, frames/s A/B), the WNMS diagnostic bits,
# the fault reproducer of the GPU tier.   gpurun -- 'bash tools/exp/r6a.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/r6a; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "frames/s (", round(d["value_min"],1), "-", round(d["value_max"],1), ") repeats", d["repeats"], "timed_gpu_s", d["timed_gpu_s"], "graph", d["config"]["hip_graph"], "digest", d["config"]["results_sha256_all_steps"], "inflight", d["config"]["batches_in_flight_per_gpu"])'
timeout -s KILL 1500 python -m pytest tests/test_build.py tests/test_kernels.py tests/test_dist.py -m gpu -q -x -k "fault or wnms_chunked or block64 or block_fusion or rccl_gather or atan2f" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for i in 1 2; do
  for v in "" "--graph" "--graph --inflight 2" "--inflight 2"; do
    echo "[$v]  $(timeout -s KILL 300 python bench.py --steps 60 --warmup 5 --repeats 5 --no-cpu-baseline --backbone-reps 0 $v 2>$O/err.txt | tail -1 | tee "$O/bench_$(echo $v | tr -d ' -')_$i.json" | python -c "$P")"
  done
done | tee $O/ab.txt
timeout -s KILL 300 python tools/enqueue_time.py --procs 1 --steps 30 2>&1 | grep process | tee $O/enqueue.txt
timeout -s KILL 300 python tools/enqueue_time.py --procs 1 --steps 30 --graph 2>&1 | grep process | tee -a $O/enqueue.txt
timeout -s KILL 600 python tools/enqueue_time.py --procs 8 --steps 30 --graph 2>&1 | grep process | tee -a $O/enqueue.txt
tail -3 $O/err.txt
