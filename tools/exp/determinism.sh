#!/bin/bash
# Run-to-run determinism soak (run THROUGH gpurun): the digest over EVERY harvested step's host results (bench.py
# config.results_sha256_all_steps) of repeated identical runs, three batches in flight.   determinism.sh [RUNS=3] [STEPS=100]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/det; mkdir -p $O; RUNS=${1:-3}; STEPS=${2:-100}
P='import sys,json; d=json.loads(sys.stdin.read()); c=d["config"]; print(c["results_sha256_all_steps"], c["results_sha256_last_step"], round(d["value"],1), "frames/s", d["steps"]*d["repeats"]+d["warmup"], "steps harvested")'
{
for cfg in "" "--inflight 1" "--inflight 2" "--inflight 4" "--dtype f16" "--config kitti --dtype bf16" "--config kitti --dtype f16"; do
  echo "== bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --backbone-reps 0 $cfg   ($RUNS runs: all-steps digest, last-step digest)"
  for r in $(seq $RUNS); do timeout -s KILL 300 python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --backbone-reps 0 $cfg 2>/dev/null | tail -1 | python -c "$P"; done
done
} > $O/determinism.txt 2>&1
cat $O/determinism.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "independent_of_block_fusion" 2>&1 | tail -2
