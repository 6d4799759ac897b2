# round 3, batch zm: NMS clip with the sorted edge list as an index register (corners + angles in LDS: 6 KB / wave instead of 10)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3zm; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_kernels.py tests/test_graph.py -m gpu -q -x -k "wnms or pair_overlap or pipeline or postprocess or evaluate" 2>&1 | tail -2
for l in base new base new; do echo "$l $(if [ $l = base ]; then export RANGEDET_HIP_LIB=$PWD/rangedet_amd/librangedet_hip_base.so; fi; python tools/wnms_bench.py 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*kept)//')"; done | tee $O/nms.txt
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["kernel_ms_per_frame"]["wnms"],4))'; }
for i in 1 2 3; do echo "base $(RANGEDET_HIP_LIB=$PWD/rangedet_amd/librangedet_hip_base.so b)"; echo "new  $(b)"; done | tee $O/ab.txt
