# round 3, batch zb: four-wave scan with grouped staging (default) vs the single-wave scan (RD_WNMS_SCAN1=1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3zb; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_kernels.py tests/test_graph.py -m gpu -q -x -k 'wnms or pair_overlap or postprocess or pipeline or evaluate' 2>&1 | tail -2
for s in 1 0; do echo "SCAN1=$s $( if [ $s = 1 ]; then export RD_WNMS_SCAN1=1; fi; python tools/wnms_bench.py 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*kept)//')"; done | tee $O/nms.txt
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["kernel_ms_per_frame"]["wnms"],4))'; }
for i in 1 2 3; do echo "scan1 $(RD_WNMS_SCAN1=1 b)"; echo "scan4 $(b)"; done | tee $O/ab.txt
