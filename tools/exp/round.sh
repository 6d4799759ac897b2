#!/bin/bash
# Round-end calls on the GPU box (run THROUGH gpurun), one script instead of a one-off per call:
#   round.sh final TAG      the whole GPU tier + smoke (tools/exp/gpu_tests.sh), then the round's profile (tools/profile_round.sh TAG)
#   round.sh profile TAG    only the profile (tools/profile_round.sh TAG)
#   round.sh pmc TAG [K]    only the PMC traffic passes (re-keys profiles/TAG_pmc_traffic.json to the current source hash); K = optional
#                           pytest -k expression run first
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; mkdir -p gpurun_out/p
case "$1" in
  final)
    bash tools/exp/gpu_tests.sh
    bash tools/profile_round.sh "$2" > gpurun_out/p_round.log 2>&1
    tail -c 600 gpurun_out/p/bench.json ;;
  profile)
    bash tools/profile_round.sh "$2" > gpurun_out/p_round.log 2>&1
    tail -c 600 gpurun_out/p/bench.json ;;
  pmc)
    [ -n "$3" ] && { timeout -s KILL 600 python -m pytest tests -m gpu -q -x -k "$3" > gpurun_out/p/pytest_k.log 2>&1; tail -2 gpurun_out/p/pytest_k.log; }
    i=0; for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do i=$((i+1)); timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/p/pmc$i -- python bench.py --steps 3 --warmup 1 --repeats 1 --backbone-reps 0 --no-cpu-baseline --inflight 1 > gpurun_out/p/pmc$i.log 2>&1; done
    python tools/pmc_summary.py 8 gpurun_out/p/pmc1 gpurun_out/p/pmc2 gpurun_out/p/pmc3 > gpurun_out/p/pmc_traffic.json 2>gpurun_out/p/pmc_summary.err
    cp gpurun_out/p/pmc_traffic.json "profiles/$2_pmc_traffic.json"
    rm -rf gpurun_out/p/pmc1 gpurun_out/p/pmc2 gpurun_out/p/pmc3
    timeout -s KILL 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/p/bench_after_pmc.json
    python -c "import json;d=json.load(open('gpurun_out/p/bench_after_pmc.json'));print(round(d['value'],1),'frames/s; traffic',d['roofline']['traffic'])" ;;
  *) echo "usage: round.sh final TAG | pmc TAG [K]"; exit 2 ;;
esac
