cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3o; mkdir -p $O
(timeout -s KILL 120 python tools/conv_clock.py 2656 128 128 8; timeout -s KILL 120 python tools/conv_clock.py 2656 64 64 8) 2>&1 | grep -v amdgpu | tee $O/clock.txt
