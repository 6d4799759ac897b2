# round 3, batch d: what bounds the 128->128 convs on the 8 x 30 tiles (ablation build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3d; mkdir -p $O
DEV=$GRAFT_REPO_ROOT/rangedet_amd/librangedet_hip_dev.so
for d in 0 16 8 4 2 32; do RANGEDET_HIP_LIB=$DEV RD_CONV3_DBG=$d C128=1 WS=2656,664 timeout -s KILL 120 python tools/conv64_bench.py | grep "128->128"; done 2>&1 | grep -v amdgpu.ids | tee $O/conv128_dbg.txt
