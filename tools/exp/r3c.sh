# round 3, batch c: DMA roles split by wave (RD_CONV_ROLE: bit 0 cout 64, bit 1 cout 128)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3c; mkdir -p $O
for r in 0 3; do RD_CONV_ROLE=$r timeout -s KILL 900 python -m pytest tests/test_kernels.py -m gpu -q -x -k "conv" 2>&1 | tail -1; done
for r in 0 1; do for hb in 0 1; do echo "ROLE=$r"; RD_CONV_ROLE=$r RD_CONV_HB3=$hb C128=1 timeout -s KILL 120 python tools/conv64_bench.py; done; done 2>&1 | grep -v amdgpu.ids | tee $O/conv64.txt
for r in 0 2; do echo "ROLE=$r"; RD_CONV_ROLE=$r C128=1 WS=2656,664,166 timeout -s KILL 120 python tools/conv64_bench.py | grep "128->128"; done 2>&1 | grep -v amdgpu.ids | tee -a $O/conv64.txt
for i in 1 2; do for r in 0 1 3; do echo "ROLE=$r $(RD_CONV_ROLE=$r timeout -s KILL 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["meta_dla_forward"]["frac_hbm_peak"],4), round(d["roofline"]["frac"],4))')"; done; done | tee $O/ab.txt
RD_CONV_ROLE=3 timeout -s KILL 600 python -m pytest tests/test_graph.py -m gpu -q -x -k "bf16 or e2e" 2>&1 | tail -1
