# round 4: the whole GPU tier + smoke + the bench line (new kernel-alone replay timing)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4f
timeout -s KILL 1500 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r4f/pytest_gpu.log 2>&1; tail -14 gpurun_out/r4f/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -2
timeout -s KILL 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4f/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4f/bench.json').read()); r=d['roofline']
print(round(d['value'],1), d['meta_dla_forward']['frac_hbm_peak'], {k:(round(r[k],4) if isinstance(r[k],float) else r[k]) for k in ('frac','achieved','avg_launch_ms','launches_per_step','serial_ms_per_step','ms_per_step','serial_forward_ms','event_overhead_factor','in_step_launch_ms','achieved_in_step')})
print(d['kernel_ms_per_frame']); print({k:d['meta_kernel'][k] for k in ('frac','avg_launch_ms','frac_mfma_peak')})
PY
