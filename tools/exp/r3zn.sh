cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3zn; mkdir -p $O
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline --wnms-cap "$1" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["kernel_ms_per_frame"]["wnms"],4))'; }
for i in 1 2 3; do echo "cap 8192 $(b 8192)"; echo "cap 4096 $(b 4096)"; echo "cap 2048 $(b 2048)"; done | tee $O/ab.txt
