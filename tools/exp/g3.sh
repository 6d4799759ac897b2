# round 5, call 3: parity of the fused block on the GPU (kernel test, all three production-geometry cases, the e2e graph tests)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/g3
(timeout -s KILL 1500 python -m pytest tests -m gpu -q -k "block64 or production_launch or e2e_bf16_tolerance or e2e_small or kitti_full_size or evaluate_loop" --durations=5 > gpurun_out/g3/pytest.txt 2>&1; tail -12 gpurun_out/g3/pytest.txt)
grep -n "fused block\|distinct launch forms" gpurun_out/g3/pytest.txt | head -30
