# round 3, batch t: cls + reg tower convs as one launch per level-layer (RD_NO_PAIR=1: 24 launches instead of 12), end to end
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3t; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_graph.py -m gpu -q -x 2>&1 | tail -3
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["meta_dla_forward"]["frac_hbm_peak"],4), round(d["roofline"]["frac"],4), d["roofline"]["launches_per_step"])'; }
for i in 1 2 3; do echo "NO_PAIR $(RD_NO_PAIR=1 b)"; echo "PAIR    $(b)"; done | tee $O/ab.txt
