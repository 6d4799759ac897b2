# round 4: WNMS rejection test -- tests, the NMS chain alone (A/B), bench A/B, and the new bench line incl. the CPU thread sweep
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4d
timeout -s KILL 600 python -m pytest tests/test_kernels.py tests/test_graph.py -m gpu -x -q -s -k "wnms or spurious or overlap or postprocess or tie_order or pipeline or full_size_f32 or e2e_bf16_tolerance or interleaved" > gpurun_out/r4d/tests.log 2>&1; tail -4 gpurun_out/r4d/tests.log; grep -E "spurious golden|far pairs" gpurun_out/r4d/tests.log
RD_WNMS_NO_SKIP=1 timeout -s KILL 200 python tools/wnms_bench.py 2>/dev/null | tail -1
timeout -s KILL 200 python tools/wnms_bench.py 2>/dev/null | tail -1
for i in 1 2; do
RD_WNMS_NO_SKIP=1 timeout -s KILL 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4d/bench_noskip_$i.json
timeout -s KILL 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4d/bench_skip_$i.json
done
timeout -s KILL 600 python bench.py --steps 40 --warmup 5 2>gpurun_out/r4d/bench_full.err | tail -1 > gpurun_out/r4d/bench_full.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4d/bench_*.json')):
    d=json.loads(open(f).read()); print(f, round(d['value'],1), d['meta_dla_forward']['frac_hbm_peak'], d['kernel_ms_per_frame']['wnms'])
d=json.loads(open('gpurun_out/r4d/bench_full.json').read())
print(json.dumps(d['cpu_baseline'])[:600]); r=d['roofline']; print({k:r[k] for k in ('frac','avg_launch_ms','launches_per_step','serial_ms_per_step','ms_per_step','serial_le_step')})
PY
