# round 3, batch zq: what the NMS chain costs end to end now (bench with the NMS kernels not enqueued vs the default)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3zq; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],3))'
for i in 1 2 3; do
echo "with NMS    $(python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P")"
echo "without NMS $(python -c 'import sys; sys.argv=["bench.py","--steps","60","--warmup","5","--no-cpu-baseline"]; import rangedet_amd.pipeline as P; P.BatchPostProcessor.enqueue_nms=lambda self, stream=None: None; import bench; bench.main()' 2>/dev/null | tail -1 | python -c "$P")"
done | tee $O/gap.txt
