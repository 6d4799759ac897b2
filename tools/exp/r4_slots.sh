#!/bin/bash
# round 4: the two-workgroup conv forms launched with ONE workgroup per CU (RD_CONV_SLOTS=1), so that the launches of two (or more)
# streams co-reside on every CU -- an HBM-bound 64-channel layer of one batch beside an MFMA-bound tower layer of the other
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/r4s; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "frames/s  ms/step", round(d["ms_per_step"],3))'
X="RANGEDET_HIP_LIB=rangedet_amd/librangedet_hip_slots.so"
for rep in 1 2; do
  for cfg in "|2" "$X RD_CONV_SLOTS=1|2" "$X RD_CONV_SLOTS=1|3" "$X RD_CONV_SLOTS=1|4" "$X|2"; do
    E="${cfg%%|*}"; N="${cfg##*|}"
    echo "[$E] inflight $N: $(env $E timeout -s KILL 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --inflight $N 2>/dev/null | tail -1 | python -c "$P")"
  done
done | tee $O/ab.txt
