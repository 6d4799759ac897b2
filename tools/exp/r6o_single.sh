#!/bin/bash
# round 6, third session: ONE pipeline alone (--inflight 1): post-processing on a side stream (its wait parked for the whole forward) or on the launch stream
# (profiles/r06o_single_pipeline_post_stream_ab.txt was taken when the side stream was still the default of a pipeline alone)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "frames/s", d["ms_per_step"], d["config"]["results_sha256_all_steps"])'
for rep in 1 2 3; do
  echo -n "--inflight 1, side stream (RD_POST_SIDE_STREAM=1): "; RD_DEV_SWITCHES=1 RD_POST_SIDE_STREAM=1 timeout -s KILL 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --backbone-reps 0 --min-timed-s 5 --inflight 1 2>/dev/null | tail -1 | python -c "$P"
  echo -n "--inflight 1, launch stream (default since this A/B): "; timeout -s KILL 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --backbone-reps 0 --min-timed-s 5 --inflight 1 2>/dev/null | tail -1 | python -c "$P"
done
