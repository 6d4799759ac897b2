#!/bin/bash
# round 6, third session: the default bench line of the final tree, and the multi-GPU code path under the driver's launcher form on one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; mkdir -p gpurun_out/p
python bench.py > gpurun_out/p/bench_final.json 2>/dev/null
RD_BENCH_GATHER=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --backbone-reps 0 2>gpurun_out/p/torchrun_err.txt | tail -1 > gpurun_out/p/bench_torchrun_gather.json
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --backbone-reps 0 2>/dev/null | tail -1 > gpurun_out/p/bench_plain_steps20.json
python - <<'P'
import json
for f in ("bench_final", "bench_torchrun_gather", "bench_plain_steps20"):
    d = json.loads(open("gpurun_out/p/%s.json" % f).read().strip().splitlines()[-1]); c = d["config"]
    print(f, round(d["value"], 1), d["ms_per_step"], c["launcher"], c["rccl_version"], c["gather_matches_local"], c["results_sha256_all_steps"], d.get("region_ms_by_rank"))
P
tail -3 gpurun_out/p/torchrun_err.txt
