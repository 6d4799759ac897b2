# round 3, batch zk: epilogue-only kernel arguments re-read from the kernarg segment per tile (fewer spilled SGPRs in the MFMA phase)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3zk; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_kernels.py -m gpu -q -x -k "conv or deconv" 2>&1 | tail -1
for l in base new; do echo "$l $(if [ $l = base ]; then export RANGEDET_HIP_LIB=$PWD/rangedet_amd/librangedet_hip_base.so; fi; C128=1 WS=2656,664 python tools/conv64_bench.py 2>&1 | grep -v amdgpu | tr '\n' ';')"; done | tee $O/conv.txt
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["meta_dla_forward"]["frac_hbm_peak"],4), round(d["roofline"]["frac"],4))'; }
for i in 1 2 3; do echo "base $(RANGEDET_HIP_LIB=$PWD/rangedet_amd/librangedet_hip_base.so b)"; echo "new  $(b)"; done | tee $O/ab.txt
