# round 3, batch z: weighted-NMS pair kernel with lane-private candidate lists, tile width 32 / 16 / 8 (RD_WNMS_CT)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3z; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_kernels.py -m gpu -q -x -k "wnms or pair_overlap" 2>&1 | tail -2
for ct in 32 16 8; do echo "CT=$ct $(RD_WNMS_CT=$ct python tools/wnms_bench.py 2>&1 | grep -v amdgpu | tail -1)"; done | tee $O/nms.txt
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["kernel_ms_per_frame"]["wnms"],4))'; }
for i in 1 2; do for ct in 32 16 8; do echo "CT=$ct $(RD_WNMS_CT=$ct b)"; done; done | tee $O/ab.txt
