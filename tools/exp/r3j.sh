# round 3, batch j: cout-64 layers with the weight image resident in LDS (8-wave workgroups, 16 x 30 tiles): RD_CONV_RES 0 / 1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3j; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_kernels.py -m gpu -q -x -k "conv or deconv" 2>&1 | tail -2
for r in 0 1; do echo "RES=$r"; RD_CONV_RES=$r WS=2656,1328,664 timeout -s KILL 120 python tools/conv64_bench.py; done 2>&1 | grep -v amdgpu | tee $O/conv.txt
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["meta_dla_forward"]["frac_hbm_peak"],4), round(d["roofline"]["frac"],4))'; }
for i in 1 2 3; do echo "RES=0 $(RD_CONV_RES=0 b)"; echo "RES=1 $(RD_CONV_RES=1 b)"; done | tee $O/ab.txt
timeout -s KILL 600 python -m pytest tests/test_graph.py -m gpu -q -x -k "bf16 or e2e or interleaved" 2>&1 | tail -1
