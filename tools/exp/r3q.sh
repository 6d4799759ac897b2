# round 3, batch q: proxy for grouped launches (two equal problems in one launch) + batch 16 upper bound
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3q; mkdir -p $O
timeout -s KILL 300 python tools/exp/r3q_group_proxy.py 2>&1 | grep -v amdgpu | tee $O/proxy.txt
b() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],3))'; }
for i in 1 2; do echo "batch 8  $(b --batch 8)"; echo "batch 16 $(b --batch 16 --steps 20)"; done | tee $O/batch.txt
