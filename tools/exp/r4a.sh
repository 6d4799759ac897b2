# round 4, first GPU call: the new per-step tight parity test (both types) + a baseline bench + per-step timing
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4a
timeout -s KILL 900 python -m pytest tests/test_production_layers.py -m gpu -x -q -s > gpurun_out/r4a/prod_layers.log 2>&1; tail -5 gpurun_out/r4a/prod_layers.log
timeout -s KILL 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4a/bench.json
timeout -s KILL 200 python tools/profile_steps.py bf16 5 8 > gpurun_out/r4a/steps.txt 2>&1
tail -c 600 gpurun_out/r4a/bench.json
