cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3w; mkdir -p $O
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["roofline"]["frac"],4))'; }
for i in 1 2; do echo "inflight1 NO_PAIR $(RD_NO_PAIR=1 b --inflight 1)"; echo "inflight1 PAIR    $(b --inflight 1)"; done | tee $O/ab1.txt
for i in 1 2; do echo "inflight3 NO_PAIR $(RD_NO_PAIR=1 b --inflight 3)"; echo "inflight3 PAIR    $(b --inflight 3)"; done | tee $O/ab3.txt
