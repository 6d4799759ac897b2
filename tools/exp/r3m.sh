# round 3, batch m: one k-step per tap for a last chunk with <= 16 real channels (RD_CONV_HALF 0 / 1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3m; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_kernels.py -m gpu -q -x -k "conv or deconv" 2>&1 | tail -1
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["meta_dla_forward"]["frac_hbm_peak"],4), round(d["roofline"]["frac"],4))'; }
for i in 1 2 3; do echo "HALF=0 $(RD_CONV_HALF=0 b)"; echo "HALF=1 $(RD_CONV_HALF=1 b)"; done | tee $O/ab.txt
for h in 0 1; do RD_CONV_HALF=$h timeout -s KILL 200 python tools/profile_steps.py bf16 5 8 2>/dev/null | grep -E "res1_unit1_conv1|conv_0_lvl_0|sum of"; done | tee $O/steps.txt
timeout -s KILL 600 python -m pytest tests/test_graph.py -m gpu -q -x -k "bf16 or e2e or kitti" 2>&1 | tail -1
