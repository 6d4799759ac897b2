# Round profile (run on the GPU box through gpurun): bench line, rocprofv3 kernel-trace stats with one and two batches in
# flight, PMC passes (separate --pmc runs, kernel-trace only), per-plan-step timing.  Results under gpurun_out/p; copy the
# summaries into profiles/ (see profiles/README.md).   usage: bash tools/profile_round.sh [tag, e.g. r04c]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/p
# (--backbone-reps 0 in the rocprofv3 / PMC passes: no stand-alone backbone replay, so every profiled launch belongs to a full forward and
#  the per-kernel means are those of the bench's `roofline` block)
# per-kernel durations: one batch in flight (kernels of two batches overlapping would inflate each other's durations; the
# bench's roofline block times its kernels in a serial replay on one stream, which is what this pass must agree with)
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p/stats -- python bench.py --steps 10 --warmup 3 --repeats 1 --backbone-reps 0 --no-cpu-baseline --inflight 1 > gpurun_out/p/stats.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p/stats2 -- python bench.py --steps 10 --warmup 3 --repeats 1 --backbone-reps 0 --no-cpu-baseline > gpurun_out/p/stats2.log 2>&1
i=0; for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do i=$((i+1)); timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/p/pmc$i -- python bench.py --steps 3 --warmup 1 --repeats 1 --backbone-reps 0 --no-cpu-baseline --inflight 1 > gpurun_out/p/pmc$i.log 2>&1; done
python tools/pmc_summary.py 8 gpurun_out/p/pmc1 gpurun_out/p/pmc2 gpurun_out/p/pmc3 > gpurun_out/p/pmc_traffic.json 2>gpurun_out/p/pmc_summary.err
# (the bench line is taken AFTER the PMC passes so that it can quote their traffic: `profile_round.sh r04c` puts the summary where
#  bench.py looks for it -- profiles/<tag>_pmc_traffic.json, keyed on the hash of the kernel sources)
[ -n "$1" ] && cp gpurun_out/p/pmc_traffic.json profiles/$1_pmc_traffic.json
timeout -s KILL 400 python bench.py --steps 100 --warmup 5 2>/dev/null | tail -1 > gpurun_out/p/bench.json
timeout -s KILL 200 python tools/profile_steps.py bf16 3 8 > gpurun_out/p/steps.txt 2>&1
for d in stats stats2; do f=$(find gpurun_out/p/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/p/${d}_kernel_stats.csv; done
rm -rf gpurun_out/p/pmc1 gpurun_out/p/pmc2 gpurun_out/p/pmc3 gpurun_out/p/stats gpurun_out/p/stats2
# the other arithmetic type and the KITTI-shaped two-class configuration (BASELINE configs[4])
timeout -s KILL 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --dtype f16 2>/dev/null | tail -1 > gpurun_out/p/bench_waymo_f16.json
timeout -s KILL 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --config kitti --dtype bf16 2>/dev/null | tail -1 > gpurun_out/p/bench_kitti_bf16.json
timeout -s KILL 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --config kitti --dtype f16 2>/dev/null | tail -1 > gpurun_out/p/bench_kitti_f16.json
bash tools/pmc_meta.sh > gpurun_out/p/meta_pmc.log 2>&1   # Meta-Kernel instruction mix -> gpurun_out/p/meta_pmc.json
ls -la gpurun_out/p; tail -c 400 gpurun_out/p/bench.json
