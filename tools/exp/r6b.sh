#!/bin/bash
# round 6, experiments of VERDICT r5 items 3 / 7 that need no new kernel:
#   (1) RD_PAIR=2: the cls and reg tower conv_0 of a level -- the two tower convs that read the SAME tensor -- as one two-problem launch
#   (2) frames per launch: bench --batch 4 / 8 / 16 / 24 (what co-scheduling the small W = 166 / 332 launches of several batches could give)
# gpurun -- 'bash tools/exp/r6b.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/r6b; mkdir -p $O
bash tools/exp/ab.sh r6b_pair "RD_PAIR=2" "" 3 both 2>&1 | tee $O/pair.txt
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "frames/s  ms/step", round(d["ms_per_step"],3), " frames/step", d["config"]["frames_per_step"], " inflight", d["config"]["batches_in_flight_per_gpu"])'
for i in 1 2; do
  for b in 4 8 16 24; do
    for f in 3 2; do
      echo "batch $b inflight $f: $(timeout -s KILL 300 python bench.py --batch $b --inflight $f --steps $((320 / b)) --warmup 4 --repeats 5 --no-cpu-baseline --backbone-reps 0 2>$O/err.txt | tail -1 | python -c "$P")"
    done
  done
done | tee $O/batch.txt
# (3) per-launch roofline table: FLOPs against the MFMA roof AND the launch form's bytes against the HBM roof (tools/profile_steps.py)
timeout -s KILL 300 python tools/profile_steps.py bf16 5 8 > $O/steps_roofline.txt 2>&1; tail -3 $O/steps_roofline.txt
