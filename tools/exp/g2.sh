# round 5, call 2: the fused BasicBlock in the product -- parity (kernel test, production-geometry test), then the A/B against the two launches
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/g2
(timeout -s KILL 900 python -m pytest tests -m gpu -x -q -k "block64 or (production_launch and bf16) or e2e_bf16_tolerance or full_size_bf16 or interleaved" > gpurun_out/g2/pytest.txt 2>&1; tail -6 gpurun_out/g2/pytest.txt)
bash tools/exp/ab.sh g2 "RD_NO_FUSE_BLOCK=1" "" 3 both
