# the whole GPU tier (what the driver runs at round end) + smoke
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/t
timeout -s KILL 1700 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/t/pytest_gpu.log 2>&1; tail -16 gpurun_out/t/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -2
