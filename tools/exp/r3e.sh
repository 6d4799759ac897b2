# round 3, batch e: fp16 mode -- parity tests on the GPU, bench in both 16-bit types
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3e; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_kernels.py -m gpu -q -x -k "conv or meta or head or deconv" 2>&1 | tail -15
timeout -s KILL 900 python -m pytest tests/test_graph.py -m gpu -q -x -s -k "e2e_bf16_tolerance" 2>&1 | grep -E "oracle|passed|failed|Error" | tail -12
for dt in bf16 f16 bf16 f16; do echo "$dt $(timeout -s KILL 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --dtype $dt 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["meta_dla_forward"]["frac_hbm_peak"],4), round(d["roofline"]["frac"],4), d["config"]["wnms_candidates"], d["config"]["wnms_kept"])')"; done | tee $O/ab.txt
