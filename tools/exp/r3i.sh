# round 3, batch i: incremental tile decode (no integer divisions per unit) A/B old vs new library; W8 on top
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3i; mkdir -p $O
OLD=$GRAFT_REPO_ROOT/rangedet_amd/librangedet_hip_old.so
timeout -s KILL 900 python -m pytest tests/test_kernels.py -m gpu -q -x -k "conv or deconv" 2>&1 | tail -1
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["meta_dla_forward"]["frac_hbm_peak"],4), round(d["roofline"]["frac"],4))'; }
for i in 1 2 3; do
  echo "old $(RANGEDET_HIP_LIB=$OLD b)"; echo "new $(b)"; echo "new+W8 $(RD_CONV_W8=1 b)"
done | tee $O/ab.txt
C128=1 WS=2656,664,166 timeout -s KILL 120 python tools/conv64_bench.py 2>&1 | grep -v amdgpu | tee $O/conv_new.txt
