# round 3, batch s: the two-problem launch (cls + reg tower conv of a level): parity on the GPU + microbench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3s; mkdir -p $O
timeout -s KILL 600 python -m pytest tests/test_kernels.py -m gpu -q -x -k "conv_pair" 2>&1 | tail -3
timeout -s KILL 300 python tools/exp/r3s_pair_bench.py 2>&1 | grep -v amdgpu | tee $O/pair.txt
