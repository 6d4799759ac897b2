# round 3, batch ze: finer launches -- 4 frames per step with 3 / 4 batches in flight vs the default 8 x 2
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3ze; mkdir -p $O
b() { python bench.py --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],3))'; }
for i in 1 2; do echo "8x2 $(b --steps 60)"; echo "4x4 $(b --steps 120 --batch 4 --inflight 4)"; echo "4x3 $(b --steps 120 --batch 4 --inflight 3)"; echo "4x2 $(b --steps 120 --batch 4 --inflight 2)"; echo "6x3 $(b --steps 80 --batch 6 --inflight 3)"; done | tee $O/ab.txt
