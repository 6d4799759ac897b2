#!/bin/bash
# A/B of two builds on the KITTI-shaped two-class configuration and in fp16 (RANGEDET_HIP_LIB = the other build)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/r4q; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "frames/s")'
for rep in 1 2 3; do
  for args in "--config kitti --dtype bf16" "--config kitti --dtype f16" "--dtype f16"; do
    for E in "RANGEDET_HIP_LIB=$1" ""; do
      echo "[$E] $args: $(env $E timeout -s KILL 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "$P")"
    done
  done
done | tee $O/ab.txt
