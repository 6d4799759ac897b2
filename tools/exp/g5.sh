cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/g5
bash tools/exp/ab.sh g5 "RD_PAIR=1" "" 2 bench
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "frames/s", d["value_min"], d["value_max"], d["ms_per_step_p5"], d["ms_per_step_p50"], d["ms_per_step_p95"])'
for n in 3 2 1; do echo "inflight $n: $(timeout -s KILL 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --inflight $n 2>/dev/null | tail -1 | python -c "$P")"; done
