# round 3, batch p: 8 x 32 tiles (34-pixel halo pitch, every MFMA column live) vs 8 x 30 (RD_CONV_WIDE 1 / 0)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3p; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_kernels.py -m gpu -q -x -k "conv or deconv" 2>&1 | tail -1
for w in 0 1; do echo "WIDE=$w"; RD_CONV_WIDE=$w C128=1 WS=2656,664 timeout -s KILL 120 python tools/conv64_bench.py; done 2>&1 | grep -v amdgpu | tee $O/conv2.txt
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["meta_dla_forward"]["frac_hbm_peak"],4), round(d["roofline"]["frac"],4))'; }
for i in 1 2; do echo "WIDE=0 $(RD_CONV_WIDE=0 b)"; echo "WIDE=1 $(RD_CONV_WIDE=1 b)"; done | tee $O/ab2.txt
