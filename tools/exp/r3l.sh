# round 3, batch l: compute with REAL data but no HBM reads (halo from an L2-resident MB) vs production, cout 64 and cout 128
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3l; mkdir -p $O
DEV=$GRAFT_REPO_ROOT/rangedet_amd/librangedet_hip_dev.so
for d in 0 256 257 16; do RANGEDET_HIP_LIB=$DEV RD_CONV3_DBG=$d C128=1 WS=2656 timeout -s KILL 120 python tools/conv64_bench.py; done 2>&1 | grep -v amdgpu.ids | tee $O/dbg256.txt
