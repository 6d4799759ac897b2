# round 3, batch zr: a middle NMS round over rows [256, R2) (RD_WNMS_R2; 0 = two rounds)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3zr; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_kernels.py tests/test_graph.py -m gpu -q -x -k "wnms or pair_overlap or pipeline or postprocess or evaluate or full_size" 2>&1 | tail -2
for r in 0 512 640 768 1024; do echo "R2=$r $(RD_WNMS_R2=$r python tools/wnms_bench.py 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*kept)//')"; done | tee $O/nms.txt
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["kernel_ms_per_frame"]["wnms"],4))'; }
for i in 1 2; do for r in 0 512 640 768 1024; do echo "R2=$r $(RD_WNMS_R2=$r b)"; done; done | tee $O/ab.txt
