# round 3, batch zg: Meta-Kernel, next tile's halo prefetch issued at tap 3 instead of at the tile start (-DRD_META_EXP=1 build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3zg; mkdir -p $O
for i in 1 2 3; do echo "new $(python tools/meta_bench.py 8 30 2>&1 | grep -v amdgpu | tail -1)"; echo "e1  $(RANGEDET_HIP_LIB=$PWD/rangedet_amd/librangedet_hip_e1.so python tools/meta_bench.py 8 30 2>&1 | grep -v amdgpu | tail -1)"; done | tee $O/meta.txt
