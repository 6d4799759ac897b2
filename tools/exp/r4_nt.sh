#!/bin/bash
# round 4: non-temporal output stores in the persistent conv (and the Meta-Kernel) -- experimental builds side by side, alternated
#   r4_nt.sh "tag1 tag2 ..." [reps]      each tag = rangedet_amd/librangedet_hip_<tag>.so ("dflt" = the shipping build)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/r4n; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "frames/s  meta+dla", round(d["meta_dla_forward"]["frac_hbm_peak"],4), " conv3 frac", round(d["roofline"]["frac"],4))'
for v in $1; do
  E=""; [ "$v" != dflt ] && E="RANGEDET_HIP_LIB=rangedet_amd/librangedet_hip_$v.so"
  env $E timeout -s KILL 200 python tools/profile_steps.py bf16 5 8 > $O/steps_$v.txt 2>&1
  echo "[$v] $(tail -1 $O/steps_$v.txt)"
done
for rep in $(seq ${2:-3}); do
  for v in $1; do
    E=""; [ "$v" != dflt ] && E="RANGEDET_HIP_LIB=rangedet_amd/librangedet_hip_$v.so"
    echo "[$v] $(env $E timeout -s KILL 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P")"
  done
done | tee $O/ab.txt
