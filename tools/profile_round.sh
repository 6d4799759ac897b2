cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/p
timeout -s KILL 300 python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/p/bench.json
# per-kernel durations: one batch in flight (kernels of two batches overlapping would inflate each other's durations; the
# bench's roofline block times its kernels in a serial replay on one stream, which is what this pass must agree with)
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p/stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --inflight 1 > gpurun_out/p/stats.log 2>&1
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p/stats2 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/p/stats2.log 2>&1
i=0; for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do i=$((i+1)); timeout -s KILL 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/p/pmc$i -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 > gpurun_out/p/pmc$i.log 2>&1; done
find gpurun_out/p -name "*.csv" | head -20; du -sh gpurun_out/p
