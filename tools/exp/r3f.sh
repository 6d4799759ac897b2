# round 3, batch f: KITTI two-class through the harness (fp16), pipeline / evaluate tests after the collect() rework, kitti bench lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3f; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_graph.py -m gpu -q -x -k "kitti or pipeline or evaluate or interleaved" 2>&1 | tail -15
timeout -s KILL 600 python -m pytest tests/test_dist.py tests/test_abi.py -q 2>&1 | tail -2
python __graft_entry__.py smoke 2>&1 | tail -3
for dt in bf16 f16; do timeout -s KILL 300 python bench.py --config kitti --dtype $dt --steps 40 --warmup 5 2>/dev/null | tail -1 > $O/bench_kitti_$dt.json; python -c "import json; d=json.load(open('$O/bench_kitti_$dt.json')); print('$dt', round(d['value'],1), d['config']['per_class'], d['config']['max_candidates_seen'])"; done
timeout -s KILL 300 python bench.py --dtype f16 --steps 60 --warmup 5 2>/dev/null | tail -1 > $O/bench_waymo_f16.json; python -c "import json; d=json.load(open('$O/bench_waymo_f16.json')); print('waymo f16', round(d['value'],1), d['cpu_baseline']['value'])"
