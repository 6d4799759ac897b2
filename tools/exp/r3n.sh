# round 3, batch n: the collective path (RCCL) with one rank: ranks_seen / rccl_version in the JSON line; full-size slow tests incl. fp16
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3n; mkdir -p $O
RD_BENCH_GATHER=1 timeout -s KILL 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>$O/gather.err | tail -1 > $O/gather.json; python -c "import json; d=json.load(open('$O/gather.json')); print(round(d['value'],1), d['config']['ranks_seen'], d['config']['rccl_version'], d['config']['gathered_frames_last_step'], d['config']['launcher'])"; tail -3 $O/gather.err
timeout -s KILL 900 python -m pytest tests/test_graph.py -m gpu -q -s -k "full_size" 2>&1 | grep -E "full size|passed|failed|kitti" | tail -12
