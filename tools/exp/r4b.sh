# round 4: all phases of a transposed conv in one launch -- unit test, production-geometry parity (bf16), per-step timing A/B, bench A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4b
timeout -s KILL 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "deconv" > gpurun_out/r4b/unit.log 2>&1; tail -3 gpurun_out/r4b/unit.log
timeout -s KILL 600 python -m pytest tests/test_production_layers.py -m gpu -x -q -s -k bf16 > gpurun_out/r4b/prod_layers.log 2>&1; tail -3 gpurun_out/r4b/prod_layers.log
timeout -s KILL 200 python tools/profile_steps.py bf16 5 8 > gpurun_out/r4b/steps_all.txt 2>&1
RD_DECONV_PER_PHASE=1 timeout -s KILL 200 python tools/profile_steps.py bf16 5 8 > gpurun_out/r4b/steps_per_phase.txt 2>&1
grep deconv gpurun_out/r4b/steps_all.txt gpurun_out/r4b/steps_per_phase.txt
for i in 1 2; do
RD_DECONV_PER_PHASE=1 timeout -s KILL 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4b/bench_per_phase_$i.json
timeout -s KILL 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4b/bench_all_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4b/bench_*.json')):
    d=json.loads(open(f).read()); print(f, round(d['value'],1), d['meta_dla_forward']['frac_hbm_peak'])
PY
