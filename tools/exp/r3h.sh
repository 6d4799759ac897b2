# round 3, batch h: 8-wave workgroups (16 x 30 tiles, shared weight ring) for the cout-128 layers (RD_CONV_W8=1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3h; mkdir -p $O
RD_CONV_W8=1 timeout -s KILL 900 python -m pytest tests/test_kernels.py -m gpu -q -x -k "conv or deconv" 2>&1 | tail -3
for w in 0 1; do echo "W8=$w"; RD_CONV_W8=$w C128=1 WS=2656,1328,664,332,166 timeout -s KILL 120 python tools/conv64_bench.py | grep "128->128"; done 2>&1 | grep -v amdgpu | tee $O/conv.txt
for i in 1 2; do for w in 0 1; do echo "W8=$w $(RD_CONV_W8=$w timeout -s KILL 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["meta_dla_forward"]["frac_hbm_peak"],4), round(d["roofline"]["frac"],4))')"; done; done | tee $O/ab.txt
RD_CONV_W8=1 timeout -s KILL 600 python -m pytest tests/test_graph.py -m gpu -q -x -k "bf16 or e2e" 2>&1 | tail -1
