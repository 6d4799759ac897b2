set -x
python bench.py --no-cpu-baseline --backbone-reps 0 --min-timed-s 5 > gpurun_out/h2d_off1.json 2> gpurun_out/h2d_err.txt
python bench.py --h2d --no-cpu-baseline --backbone-reps 0 --min-timed-s 5 > gpurun_out/h2d_on1.json 2>> gpurun_out/h2d_err.txt
python bench.py --no-cpu-baseline --backbone-reps 0 --min-timed-s 5 > gpurun_out/h2d_off2.json 2>> gpurun_out/h2d_err.txt
python bench.py --h2d --no-cpu-baseline --backbone-reps 0 --min-timed-s 5 > gpurun_out/h2d_on2.json 2>> gpurun_out/h2d_err.txt
for f in off1 on1 off2 on2; do python -c "
import json,sys
d=json.loads(open('gpurun_out/h2d_$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'], d['config']['inputs'], d['config']['results_sha256_all_steps'])"; done
tail -3 gpurun_out/h2d_err.txt
