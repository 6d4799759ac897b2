cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/g1
(timeout -s KILL 600 python tools/micro/block_dev.py gpu bench > gpurun_out/g1/block.txt 2>&1; tail -12 gpurun_out/g1/block.txt)
(timeout -s KILL 900 python -m pytest tests -m gpu -x -q -k "rccl or spurious or pair_overlap or wnms_golden or kitti_pipeline_two_class or pipeline_postprocess or box_formats or score_filter" > gpurun_out/g1/pytest.txt 2>&1; tail -8 gpurun_out/g1/pytest.txt)
(timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/g1/bench.json; python -c "
import json;d=json.load(open('gpurun_out/g1/bench.json'));print(d['value'],d['value_min'],d['value_max'],d['ms_per_step_p5'],d['ms_per_step_p50'],d['ms_per_step_p95'],d['meta_dla_forward']['frac_hbm_peak'],d['roofline']['frac'])")
bash tools/pmc_conv_classes.sh > gpurun_out/g1/pcc.txt 2>&1; tail -5 gpurun_out/g1/pcc.txt
