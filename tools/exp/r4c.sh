# round 4: virtual concat (two-tensor conv) + all-phase deconv + relu->add epilogue: tests, per-step timing, bench A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4c
timeout -s KILL 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "deconv or cat_two or conv3x3_ex or conv2d_bn" > gpurun_out/r4c/unit.log 2>&1; tail -3 gpurun_out/r4c/unit.log
timeout -s KILL 900 python -m pytest tests/test_production_layers.py tests/test_graph.py -m gpu -x -q -s -k "production or e2e_bf16_tolerance" > gpurun_out/r4c/prod_layers.log 2>&1; tail -3 gpurun_out/r4c/prod_layers.log
timeout -s KILL 200 python tools/profile_steps.py bf16 5 8 > gpurun_out/r4c/steps.txt 2>&1
RD_CONCAT_BUFFER=1 timeout -s KILL 200 python tools/profile_steps.py bf16 5 8 > gpurun_out/r4c/steps_concat_buffer.txt 2>&1
for i in 1 2; do
RD_CONCAT_BUFFER=1 RD_DECONV_PER_PHASE=1 timeout -s KILL 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4c/bench_old_$i.json
timeout -s KILL 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4c/bench_new_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c/bench_*.json')):
    d=json.loads(open(f).read()); print(f, round(d['value'],1), d['meta_dla_forward']['frac_hbm_peak'])
PY
