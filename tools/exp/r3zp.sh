# round 3, batch zp: the first-round alive list written by the four-wave scan itself (one launch fewer on the NMS chain)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3zp; mkdir -p $O
for l in base new base new; do echo "$l $(if [ $l = base ]; then export RANGEDET_HIP_LIB=$PWD/rangedet_amd/librangedet_hip_base.so; fi; python tools/wnms_bench.py 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*kept)//')"; done | tee $O/nms.txt
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["kernel_ms_per_frame"]["wnms"],4))'; }
for i in 1 2 3 4; do echo "base $(RANGEDET_HIP_LIB=$PWD/rangedet_amd/librangedet_hip_base.so b)"; echo "new  $(b)"; done | tee $O/ab.txt
