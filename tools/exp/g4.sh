# round 5, call 4: the first block fused too -- parity, then A/B (first block as two launches vs one)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/g4
(timeout -s KILL 1200 python -m pytest tests -m gpu -x -q -k "block64 or (production_launch and (bf16 or kitti)) or e2e_bf16_tolerance or kitti_pipeline or full_size_bf16" > gpurun_out/g4/pytest.txt 2>&1; tail -5 gpurun_out/g4/pytest.txt)
bash tools/exp/ab.sh g4 "RD_NO_FUSE_FIRST=1" "" 3 both
