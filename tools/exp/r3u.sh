cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3u; mkdir -p $O
timeout -s KILL 300 python tools/exp/r3u_grp_overhead.py 2>&1 | grep -v amdgpu | tee $O/overhead.txt
