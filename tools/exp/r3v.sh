cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3v; mkdir -p $O
timeout -s KILL 600 python -m pytest tests/test_kernels.py -m gpu -q -x -k "conv_pair" 2>&1 | tail -2
timeout -s KILL 300 python tools/exp/r3u_grp_overhead.py 2>&1 | grep -v amdgpu | tee $O/overhead.txt
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["meta_dla_forward"]["frac_hbm_peak"],4), round(d["roofline"]["frac"],4), d["roofline"]["launches_per_step"])'; }
for i in 1 2 3; do echo "NO_PAIR $(RD_NO_PAIR=1 b)"; echo "PAIR    $(b)"; done | tee $O/ab.txt
