#!/bin/bash
# One parametrised A/B runner for the GPU box (run it THROUGH gpurun: `gpurun -- 'bash tools/exp/ab.sh OUT "ENV_A" "ENV_B" ...'`).
# Every performance experiment of rounds 3 and 4 (profiles/EXPERIMENTS.md) is an instance of it: two settings of the library's
# development switches (environment variables) -- or two builds, RANGEDET_HIP_LIB=... -- alternated on ONE box
# inside ONE call, because boxes differ by +-2 %.
#
#   ab.sh OUT "ENV_A" "ENV_B" [REPS=2] [MODE=bench|steps|both] [PYTEST_K]
#     OUT       sub-directory of gpurun_out/
#     ENV_A/B   e.g. "RD_CONV_BODY=0" and "" (empty = defaults); several assignments separated by spaces
#     MODE      bench: frames/s + meta_dla frac + conv3 frac per run; steps: tools/profile_steps.py per setting; both
#     PYTEST_K  optional -k expression: the matching GPU tests run first (parity before speed)
#
# examples (the round-4 experiments):
#   ab.sh r4b "RD_DECONV_PER_PHASE=1" ""  2 both deconv        all phases of a transposed conv in one launch
#   ab.sh r4c "RD_CONCAT_BUFFER=1" ""     2 both cat_two       the concat never materialised
#   ab.sh r4d "RD_WNMS_NO_SKIP=1" ""      2 bench wnms         the weighted NMS's rejection test
#   ab.sh r4g "RD_CONV_BODY=0" ""         2 both conv3x3_ex    heterogeneous tile bodies
# Round 6: the release library and the release lowering take no switches.  This runner sets RD_DEV_SWITCHES=1 (Python-side switches:
# rangedet_amd/devswitch.py) and, when tools has built it (`python -m rangedet_amd.build --dev` before the gpurun call), loads the
# -DRD_DEV_SWITCHES library so that the native RD_CONV_* / RD_WNMS_* variables work too.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp RD_DEV_SWITCHES=1
[ -z "$RANGEDET_HIP_LIB" ] && [ -f rangedet_amd/librangedet_hip_dev.so ] && export RANGEDET_HIP_LIB="$PWD/rangedet_amd/librangedet_hip_dev.so"
O=gpurun_out/${1:?out dir}; A="$2"; B="$3"; REPS=${4:-2}; MODE=${5:-bench}; K="$6"
mkdir -p "$O"
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "frames/s  meta+dla frac", round(d["meta_dla_forward"]["frac_hbm_peak"],4), " conv3 frac", round(d["roofline"]["frac"],4), " wnms ms/frame", round(d["kernel_ms_per_frame"]["wnms"],4))'
if [ -n "$K" ]; then
  timeout -s KILL 900 python -m pytest tests -m gpu -q -x -k "$K" > "$O/pytest.log" 2>&1; tail -3 "$O/pytest.log"
fi
if [ "$MODE" = steps ] || [ "$MODE" = both ]; then
  env $A timeout -s KILL 200 python tools/profile_steps.py bf16 5 8 > "$O/steps_A.txt" 2>&1
  env $B timeout -s KILL 200 python tools/profile_steps.py bf16 5 8 > "$O/steps_B.txt" 2>&1
  echo "A: [$A] $(tail -1 "$O/steps_A.txt")"; echo "B: [$B] $(tail -1 "$O/steps_B.txt")"
  # the steps whose time differs by more than 3 us
  paste <(awk '$2 ~ /conv|deconv|meta|nchw|sorted|concat/ {print $3, $(NF-18)}' "$O/steps_A.txt" 2>/dev/null) <(awk '$2 ~ /conv|deconv|meta|nchw|sorted|concat/ {print $3, $(NF-18)}' "$O/steps_B.txt" 2>/dev/null) |
    awk '{d=$4-$2; if (d>3||d<-3) print}' | head -40
fi
if [ "$MODE" = bench ] || [ "$MODE" = both ]; then
  for i in $(seq "$REPS"); do
    echo "A [$A]  $(env $A timeout -s KILL 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | tee "$O/bench_A_$i.json" | python -c "$P")"
    echo "B [$B]  $(env $B timeout -s KILL 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | tee "$O/bench_B_$i.json" | python -c "$P")"
  done | tee "$O/ab.txt"
fi
