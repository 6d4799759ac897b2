# round 3, batch zo: rows of the first NMS round (256 in the tree) -- builds with 128 / 192 / 384 / 512
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3zo; mkdir -p $O
for r in 256 128 192 384 512; do echo "R1=$r $( if [ $r != 256 ]; then export RANGEDET_HIP_LIB=$PWD/rangedet_amd/librangedet_hip_r$r.so; fi; python tools/wnms_bench.py 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*kept)//')"; done | tee $O/nms.txt
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["kernel_ms_per_frame"]["wnms"],4))'; }
for i in 1 2; do for r in 256 128 192 384 512; do echo "R1=$r $( if [ $r != 256 ]; then export RANGEDET_HIP_LIB=$PWD/rangedet_amd/librangedet_hip_r$r.so; fi; b)"; done; done | tee $O/ab.txt
