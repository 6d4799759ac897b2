# round 4: residual prefetch A/B (per-step + bench), WNMS goldens on hardware, the new bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4e
timeout -s KILL 600 python -m pytest tests/test_kernels.py -m gpu -q -s -k "wnms or spurious or overlap or conv3x3_ex or deconv" > gpurun_out/r4e/tests.log 2>&1; tail -4 gpurun_out/r4e/tests.log; grep -E "spurious golden|far pairs" gpurun_out/r4e/tests.log
timeout -s KILL 600 python -m pytest tests/test_production_layers.py -m gpu -q -s -k bf16 > gpurun_out/r4e/prod.log 2>&1; tail -2 gpurun_out/r4e/prod.log
RD_CONV_PFRES=0 timeout -s KILL 200 python tools/profile_steps.py bf16 5 8 > gpurun_out/r4e/steps_nopf.txt 2>&1
timeout -s KILL 200 python tools/profile_steps.py bf16 5 8 > gpurun_out/r4e/steps_pf.txt 2>&1
paste <(grep -E "conv2|deconv" gpurun_out/r4e/steps_nopf.txt | awk '{print $3, $(NF-10), $(NF-9)}' ) <(grep -E "conv2|deconv" gpurun_out/r4e/steps_pf.txt | awk '{print $(NF-10), $(NF-9)}') | head -40
tail -1 gpurun_out/r4e/steps_nopf.txt; tail -1 gpurun_out/r4e/steps_pf.txt
for i in 1 2; do
RD_CONV_PFRES=0 timeout -s KILL 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4e/bench_nopf_$i.json
timeout -s KILL 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4e/bench_pf_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4e/bench_*.json')):
    d=json.loads(open(f).read()); r=d['roofline']; print(f, round(d['value'],1), d['meta_dla_forward']['frac_hbm_peak'], round(r['frac'],4), round(r['avg_launch_ms'],4), round(r['serial_ms_per_step'],3), round(r['ms_per_step'],3), round(r['achieved_in_step'],1))
PY
