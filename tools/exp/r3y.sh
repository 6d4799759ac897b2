cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3y; mkdir -p $O
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["roofline"]["frac"],4), d["roofline"]["launches_per_step"])'; }
for i in 1 2 3; do echo "NO_PAIR    $(RD_NO_PAIR=1 b)"; echo "PAIR<=700  $(RD_PAIR_MAXW=700 b)"; echo "PAIR<=1400 $(RD_PAIR_MAXW=1400 b)"; done | tee $O/ab.txt
