# round 3, batch za: pair kernel -- the wave's candidate pairs dealt out evenly (RD_WNMS_BAL) x tile width
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3za; mkdir -p $O
for bal in 1 0; do echo "BAL=$bal $(RD_WNMS_BAL=$bal timeout -s KILL 900 python -m pytest tests/test_kernels.py -m gpu -q -x -k 'wnms or pair_overlap' 2>&1 | tail -1)"; done
for bal in 0 1; do for ct in 8 16 32; do echo "BAL=$bal CT=$ct $(RD_WNMS_BAL=$bal RD_WNMS_CT=$ct python tools/wnms_bench.py 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*kept)//')"; done; done | tee $O/nms.txt
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["kernel_ms_per_frame"]["wnms"],4))'; }
for i in 1 2; do echo "BAL=0 CT=8 $(RD_WNMS_BAL=0 b)"; echo "BAL=1 CT=8 $(RD_WNMS_BAL=1 b)";  echo "BAL=1 CT=16 $(RD_WNMS_BAL=1 RD_WNMS_CT=16 b)"; done | tee $O/ab.txt
