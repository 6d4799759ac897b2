# round 3, batch zl: merge kernel's LDS list capacity (65 KB per single-wave workgroup by default) -- RD_WNMS_MERGE_LDS entries
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3zl; mkdir -p $O
for m in 0 4096 2048 1024 512; do echo "MERGE_LDS=$m $( if [ $m != 0 ]; then export RD_WNMS_MERGE_LDS=$m; fi; python tools/wnms_bench.py 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*kept)//')"; done | tee $O/nms.txt
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["kernel_ms_per_frame"]["wnms"],4))'; }
for i in 1 2 3; do echo "default $(b)"; echo "2048    $(RD_WNMS_MERGE_LDS=2048 b)"; echo "512     $(RD_WNMS_MERGE_LDS=512 b)"; done | tee $O/ab.txt
