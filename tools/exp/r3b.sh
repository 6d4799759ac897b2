# round 3, batch b: per-layer time per frame against frames per launch (is the Infinity Cache worth sub-batching the full-width layers?)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3b; mkdir -p $O
for b in 1 2 4 8; do timeout -s KILL 200 python tools/profile_steps.py bf16 5 $b 2>/dev/null > $O/steps_b$b.txt; tail -1 $O/steps_b$b.txt; done
for b in 4 8 16; do echo "batch=$b $(timeout -s KILL 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --batch $b 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["meta_dla_forward"]["frac_hbm_peak"],4), round(d["roofline"]["frac"],4))')"; done | tee $O/batch.txt
timeout -s KILL 600 python -m pytest tests/test_ref_python_pins.py -m gpu -q 2>&1 | tail -2
