# round 3, batch k: where the time of the resident-weight cout-64 kernel goes (ablation build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3k; mkdir -p $O
DEV=$GRAFT_REPO_ROOT/rangedet_amd/librangedet_hip_dev.so
for d in 0 256 257 16 8; do RANGEDET_HIP_LIB=$DEV RD_CONV3_DBG=$d WS=2656 timeout -s KILL 120 python tools/conv64_bench.py; done 2>&1 | grep -v amdgpu.ids | tee $O/res_dbg2.txt
