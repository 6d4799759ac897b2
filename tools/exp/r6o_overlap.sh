#!/bin/bash
# round 6, third session: kernel trace of the timed pipeline (three and four batches in flight) -> how full the GPU is (tools/overlap_trace.py)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/r6o; mkdir -p $O
for n in 3 4 2; do
  rm -rf $O/tr$n
  timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr$n -- python bench.py --steps 30 --warmup 5 --repeats 2 --backbone-reps 0 --no-cpu-baseline --inflight $n > $O/tr$n.log 2>&1
  f=$(find $O/tr$n -name "*kernel_trace.csv" | head -1)
  echo "== --inflight $n: $(tail -1 $O/tr$n.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "frames/s under the profiler")')"
  head -1 $f | cut -c1-400
  python tools/overlap_trace.py $f
  rm -rf $O/tr$n
done
