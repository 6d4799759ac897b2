# round 3, batch zf: Meta-Kernel with the centre tap's MLP folded into a constant: A/B against a build with the previous kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3zf; mkdir -p $O
for i in 1 2 3; do echo "base $(RANGEDET_HIP_LIB=$PWD/rangedet_amd/librangedet_hip_base.so python tools/meta_bench.py 8 30 2>&1 | grep -v amdgpu | tail -1)"; echo "new  $(python tools/meta_bench.py 8 30 2>&1 | grep -v amdgpu | tail -1)"; done | tee $O/meta.txt
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["meta_kernel"]["avg_launch_ms"],4), round(d["meta_kernel"]["frac"],4), round(d["meta_dla_forward"]["frac_hbm_peak"],4))'; }
for i in 1 2; do echo "base $(RANGEDET_HIP_LIB=$PWD/rangedet_amd/librangedet_hip_base.so b)"; echo "new  $(b)"; done | tee $O/ab.txt
