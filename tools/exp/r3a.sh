# round 3, experiment batch a: three halo buffers (RD_CONV_HB3) for the cout-64 layers + what the residual read costs
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3a; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_kernels.py -m gpu -q -x -k "conv" > $O/pytest_conv.log 2>&1; tail -3 $O/pytest_conv.log
DEV=$GRAFT_REPO_ROOT/rangedet_amd/librangedet_hip_dev.so
for hb in 0 1; do RD_CONV_HB3=$hb timeout -s KILL 120 python tools/conv64_bench.py; done 2>&1 | grep -v amdgpu.ids | tee $O/conv64.txt
for hb in 0 1; do for d in 64 128 16 4 8; do RANGEDET_HIP_LIB=$DEV RD_CONV_HB3=$hb RD_CONV3_DBG=$d WS=2656 timeout -s KILL 120 python tools/conv64_bench.py; done; done 2>&1 | grep -v amdgpu.ids | tee $O/conv64_dbg.txt
for i in 1 2; do for hb in 0 1; do echo "HB3=$hb $(RD_CONV_HB3=$hb timeout -s KILL 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["meta_dla_forward"]["frac_hbm_peak"],4), round(d["roofline"]["frac"],4))')"; done; done | tee $O/ab.txt
timeout -s KILL 600 python -m pytest tests/test_graph.py -m gpu -q -x -k "bf16 or e2e" > $O/pytest_graph.log 2>&1; tail -3 $O/pytest_graph.log
