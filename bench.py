#!/usr/bin/env python
"""bench.py -- range-image frames/s of the RangeDet inference hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N > 1 without RANK/WORLD_SIZE in the environment: bench.py spawns
                                                        its own N ranks, one per GPU; under torch.distributed.run it
                                                        takes the launcher's ranks)

One "step" = one full pass of the hot path over one batch (--batch, default 8) of synthetic 64x2650(pad 2656)x8 range
images per rank (BASELINE config 4 runs 64 frames over 8 GPUs = 8 per GPU; frames are independent, batching only fills
the low-resolution layers' grids):
DLA backbone + fused Meta-Kernel + 3-level heads + sigmoid/top-50000/sort + 3D box decode + score filter + weighted NMS
(config `rangedet_veh_wo_aug_4_18e`, bf16 activations/weights with fp32 accumulation; BASELINE configs[1] plus the WNMS
of configs[2]).  Inputs are resident in HBM before the timed region.  Frames shard across ranks with no data-path
collective (weak scaling); for N > 1 every step ends with the one RCCL all_gather of the padded detections
(SURVEY.md 8e), packed behind the batch's NMS and enqueued on the pipeline's communication stream (no launch stream carries it).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0
NPIX = 64 * 2656            # padded pixels of a frame (set by main() for the configuration that runs)
META_FLOP_PER_PX = 19.29e9 / (64 * 2656)   # 19.29 GFLOP per 64 x 2656 frame


def is_conv3(s):
    """Plan steps served by the persistent 3x3 bf16 kernel (csrc/k_conv3.h conv3_eligible): stride 1, and stride 2 computed as
    stride 1 with only the even columns stored (the roofline counts the ALGORITHMIC flops of the strided conv)."""
    return s["kind"] == "conv" and tuple(s["k"]) == (3, 3)


def conv_flops(plan, only_conv3=False):
    """Algorithmic FLOPs of the conv-family launches of one frame, and the number of launches (from the lowered plan)."""
    from rangedet_amd.lower import conv_steps
    fl, n = 0.0, 0
    for s, launches in conv_steps(plan.steps):   # (a "conv_pair" -- two tower convs in one launch -- counts as two convs, one launch)
        if only_conv3 and bool(s.get("in_block")) != (only_conv3 == "block"):
            continue                                 # (the two convs of a fused BasicBlock run in block64_stream_kernel: only_conv3="block")
        if s["kind"] == "conv" and (not only_conv3 or is_conv3(s)):
            fl += 2.0 * s["out"].H * s["out"].W * s["cin"] * s["cout"] * s["k"][0] * s["k"][1]
            if s.get("sc"):   # the block's 1x1 projection shortcut, accumulated in this launch's epilogue
                fl += 2.0 * s["out"].H * s["out"].W * s["sc"]["cin"] * s["cout"]
            n += launches
        elif s["kind"] == "deconv":   # bf16: every phase runs on the persistent kernel too (3x3 tap embedding)
            # every output pixel sums kh*kw/stride taps
            fl += 2.0 * s["out"].H * s["out"].W * s["cin"] * s["cout"] * s["k"][0] * s["k"][1] / s["stride_w"]
            n += launches
    return fl, int(round(n))


def conv_bytes(plan, esz, only_conv3=False):
    """Algorithmic HBM bytes of the conv-family launches of one frame: every layer reads its input once, writes its
    output once, reads its residual once (BN/ReLU/add fused), weights once (SURVEY.md section 8d bytes model)."""
    from rangedet_amd.lower import conv_steps
    by = 0.0
    for s, _ in conv_steps(plan.steps):
        if only_conv3 and bool(s.get("in_block")) != (only_conv3 == "block"):
            continue
        if s["kind"] == "deconv" or (s["kind"] == "conv" and (not only_conv3 or is_conv3(s))):
            x, o = s["x"], s["out"]
            by += (x.H * x.W * s["cin"] + o.H * o.W * s["cout"]) * esz
            if s.get("res") is not None:
                by += o.H * o.W * s["cout"] * esz
            by += s["cin"] * s["cout"] * s["k"][0] * s["k"][1] * esz
            if s.get("sc"):
                # the block's projection shortcut runs inside this launch; the ALGORITHMIC bytes stay those of the reference's
                # layer-by-layer graph: the 1x1 conv reads the block input, writes its output, and this conv
                # reads that output back as its residual
                sx = s["sc_x"]
                by += (sx.H * sx.W * s["sc"]["cin"] + 2 * o.H * o.W * s["cout"] + s["sc"]["cin"] * s["cout"]) * esz
    return by


def backbone_forward_roofline(pipe, frame, Bf, esz, reps=2):
    """north_star's target quantity: the Meta-Kernel + DLA backbone forward (every plan step before the first head conv)
    against the HBM roof.  Serial replay of those steps on the current stream, bracketed by HIP events; algorithmic bytes =
    conv-family bytes model of those layers + the Meta-Kernel's compulsory 262 B/px (1 812 MB per frame)."""
    import copy
    import torch
    steps = pipe.plan.steps
    nb = next(i for i, s in enumerate(steps) if str(s.get("name", "")).startswith("rpn_"))
    sub = copy.copy(pipe.plan)
    sub.steps = steps[:nb]
    gb = (conv_bytes(sub, esz) + NPIX * (128 * esz + 12)) / 1e9
    gf = (conv_flops(sub)[0] + META_FLOP_PER_PX * NPIX) / 1e9
    dev = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # (few repetitions on purpose: these launches are outside the pipeline replay that the `roofline` block averages, and
    # a rocprofv3 summary of this command averages over every launch of the process)
    for r in range(reps + 1):
        if r == 1:
            e0.record()
        for i in range(nb):
            pipe.exe.forward(frame, only=i, dev=dev)
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / reps / Bf
    # the Meta-Kernel alone, with nothing else resident (in the pipeline the previous batch's NMS co-runs with it)
    L = pipe.lib
    L.call("rd_prof_reset")
    L.call("rd_prof_enable", 1)
    mi = next(i for i, s in enumerate(steps) if s["kind"] == "meta")
    for r in range(10):
        pipe.exe.forward(frame, only=mi, dev=dev)
    torch.cuda.synchronize()
    mms, mcnt = L.prof()["meta"]
    L.call("rd_prof_enable", 0)
    mbytes = Bf * NPIX * (128 * esz + 12)
    meta_alone = {"avg_launch_ms": mms / max(mcnt, 1), "gbps": mbytes / (mms / max(mcnt, 1) * 1e-3) / 1e9 if mcnt else 0.0}
    meta_alone["frac_hbm_peak"] = meta_alone["gbps"] / PEAK_HBM_GBPS
    return {"meta_kernel_alone": meta_alone,"scope": "Meta-Kernel + DLA backbone forward: plan steps 0..%d (input layout, %d conv-family launches, the fused "
                     "Meta-Kernel unit), serial replay on one stream, HIP events" % (nb - 1, conv_flops(sub)[1]),
            "ms_per_frame": sec * 1e3, "algorithmic_gb_per_frame": gb, "hbm_gbps": gb / sec,
            "frac_hbm_peak": gb / sec / PEAK_HBM_GBPS, "gflop_per_frame": gf, "tflops": gf / sec / 1e3,
            "frac_mfma_peak": gf / sec / 1e3 / PEAK_BF16_TFLOPS, "target_frac_hbm_peak": 0.6}


def measured_traffic(kernel_key, batch):
    """HBM bytes per launch from the newest committed rocprofv3 PMC summary (profiles/r*_pmc_traffic.json, tools/profile_round.sh)
    that was taken on THESE kernel sources (csrc_sha16 == rangedet_amd.build.source_hash()) at this batch size; else None -- a
    figure measured on other kernels is not quoted."""
    import glob
    from rangedet_amd.build import source_hash
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
            if d.get("batch") == batch and d.get("csrc_sha16") == source_hash():
                return d[kernel_key]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
    return None


def cpu_baseline(params, frames, budget_s=30.0):
    """The oracle (PyTorch-CPU fp32 restatement + C++ decode/wnms) timed on this box's host cores: reported, not a target.
    Whole frames of the same workload until ~budget_s of CPU work is spent (at least 2 frames; the first one also pays
    torch's one-time thread-pool / allocator warm-up and is reported separately)."""
    import torch
    from oracle import graph_ref

    def one(frame):
        t0 = time.time()
        out = graph_ref.forward(frame, params)
        graph_ref.postprocess(out["fg_cls_score"][0], out["decoded_bbox"][0])
        return time.time() - t0

    # torch's intra-op pool on ALL hardware threads is not the fastest setting for these shapes (128 threads: 11.6 s per frame in
    # round 3): one frame at each of {16, 32, 64, all} threads after a warm-up frame, then the timed frames at the best setting
    ncpu = os.cpu_count() or torch.get_num_threads()
    first = one(frames[0])
    sweep = {}
    for nt in sorted({min(n, ncpu) for n in (16, 32, 64, ncpu)}):
        if sweep and nt > 64 and sweep[max(sweep)] > 1.25 * min(sweep.values()):
            break      # already past the optimum (256 threads: 116 s per frame on the round-4 box against 4.1 s at 32) -- not worth minutes
        torch.set_num_threads(nt)
        sweep[nt] = one(frames[1 % len(frames)])
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times = []
    while len(times) < 2 or (sum(times) < budget_s / 2 and len(times) < 4):
        times.append(one(frames[(2 + len(times)) % len(frames)]))
    return {"value": len(times) / sum(times), "unit": "frames/s", "cores": best, "kind": "port", "host_threads_available": ncpu,
            "first_frame_s": first, "frames_timed": len(times), "s_per_frame_by_threads": {str(k): round(v, 3) for k, v in sweep.items()},
            "sample": "%d frame(s) of the same 64x2656x8 workload at the fastest of the thread counts tried (one frame each, after a "
                      "warm-up frame): PyTorch-CPU fp32 restatement of the MXNet graph (the reference's MXNet CPU path cannot run: mxnet "
                      "is not installed) + C++ decode/wnms restatement" % len(times)}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, argv):
    """One spawned rank (python bench.py --gpus N without a launcher): the same environment torch.distributed.run would set."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    main(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` with no RANK / WORLD_SIZE: spawn the N ranks here, like the reference's eval driver starts one
    module per GPU (tools/test.py:143-161); rank 0 prints the JSON line.  A failing rank fails the whole run (mp.spawn re-raises)."""
    import torch.multiprocessing as mp
    mp.spawn(_rank_main, args=(args.gpus, _free_port(), argv), nprocs=args.gpus, join=True)


def dry_run(args, rank, world):
    """RD_BENCH_DRYRUN=1: the launch / rendezvous / gather / reporting skeleton on the host alone (gloo; no GPU, no kernels):
    every rank contributes its rank and frame list, rank 0 prints a JSON line.  Used by tests/test_dist.py."""
    import torch
    import torch.distributed as dist
    from rangedet_amd import dist as rdist
    rdist.init_process_group("gloo")
    cpus = rdist.bind_cpus(rank, world)
    shard = rdist.FrameSharding(rank, world)
    mine = torch.tensor([rank] + shard.frames_of_step(0, args.batch), dtype=torch.int64)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    # the SHIPPING gather (rangedet_amd.dist.DetectionGather: pack + one all_gather + unpack) on host buffers: every rank fills a
    # post-processor-shaped result whose rows encode (rank, frame, row), all ranks must find every frame of the step intact
    B_, cap = args.batch, 256

    class _Post:      # the fields of pipeline.BatchPostProcessor the gather reads
        pass
    post = _Post()
    post.B, post.cap = B_, cap
    rows = np.zeros((B_, cap, 12), np.float32)
    nk = np.zeros(B_, np.int32)
    for j, f in enumerate(shard.frames_of_step(3, B_)):
        nk[j] = 1 + (f * 7) % rdist.MAX_DET
        rows[j, :nk[j]] = (1000.0 * f + np.arange(nk[j], dtype=np.float32))[:, None] + 0.01 * np.arange(12, dtype=np.float32)[None, :]
    post.out, post.nkeep = rows.view(np.uint8).reshape(-1), nk.view(np.uint8)
    g = rdist.DetectionGather(post, shard, rdist.HostAlloc, rdist.HostCopyLib)
    g.enqueue()
    got = g.unpack(step=3)
    ok = sorted(got) == sorted(f for r in range(world) for f in shard.frames_of_step(3, B_, rank=r))
    for f, (rw, M) in got.items():
        m = 1 + (f * 7) % rdist.MAX_DET
        want = (1000.0 * f + np.arange(m, dtype=np.float32))[:, None] + 0.01 * np.arange(12, dtype=np.float32)[None, :]
        ok = ok and M == m and np.array_equal(rw, want)
    okt = torch.tensor([1 if ok else 0], dtype=torch.int64)
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    ident = torch.tensor(list(bytes([rank]) * 16), dtype=torch.uint8)      # (dry run: one fictitious device per rank)
    ids = [torch.zeros_like(ident) for _ in range(world)]
    dist.all_gather(ids, ident)
    rdist.check_ranks([int(a[0]) for a in allr], [bytes(i.tolist()) for i in ids], world)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": [int(a[0]) for a in allr],
                          "frames_step0": sorted(int(f) for a in allr for f in a[1:]), "max_over_ranks": float(t.item()),
                          "gather_ok": bool(okt.item()), "gathered_frames": len(got), "cpus_rank0": len(cpus) if cpus else None}), flush=True)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"],
                    help="arithmetic type of activations / weights (fp32 accumulation): bf16 = BASELINE configs[1]; f16 = the reference's "
                         "own mixed-precision type (config fp16 = True); f32 = parity mode")
    ap.add_argument("--config", default="waymo", choices=["waymo", "kitti"],
                    help="waymo = rangedet_veh_wo_aug_4_18e on 64x2650(pad 2656)x8 (BASELINE configs[1]/[2], the metric's workload); "
                         "kitti = BASELINE configs[4]: 64x2048x5 range images, vehicle + pedestrian heads (named in config.workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames", type=int, default=2, help="distinct synthetic batches cycled through")
    ap.add_argument("--inflight", type=int, default=3,
                    help="batches in flight per GPU (pipelines on separate streams).  3 since round 5: with the post-processing on each batch's own "
                         "launch stream, a third batch keeps two on the GPU while the host waits for the oldest one and enqueues the next "
                         "(960 - 972 vs 945 - 953 frames/s on one box; with side streams 3 was slower than 2)")
    ap.add_argument("--batch", type=int, default=8, help="frames per step per GPU (BASELINE config 4: 64 frames over 8 GPUs = 8 per GPU)")
    ap.add_argument("--tie-order", default="reference", choices=["reference", "stable"],
                    help="processing order of equal scores in the weighted NMS: the reference's std::sort order (default) or index order")
    ap.add_argument("--h2d", action="store_true",
                    help="PCIe-inclusive diagnostic (never the headline): every step first uploads its batch's input tensors from pinned host "
                         "memory on the batch's launch stream, instead of reading inputs resident in HBM (DESIGN.md section 6)")
    ap.add_argument("--graph", action="store_true",
                    help="hipGraph replay: every pipeline captures its ~80 launches per batch once per input set and replays them with one call "
                         "(pipeline.RangeDetPipeline(graph=True)); same results (config.results_sha256_all_steps), ~1 ms less host work per step")
    ap.add_argument("--wnms-no-skip", action="store_true",
                    help="diagnostic (rdlib.RD_WNMS_DIAG_NO_SKIP): the weighted NMS clips every pair the reference clips -- same results, for A/B and tests")
    ap.add_argument("--wnms-cap", type=int, default=8192, help="rows per frame the weighted NMS is sized for (checked every step)")
    ap.add_argument("--backbone-reps", type=int, default=10,
                    help="replays of the Meta-Kernel + DLA backbone steps for the `meta_dla_forward` block (0: skip that block -- the rocprofv3 / PMC "
                         "passes of tools/profile_round.sh use 0 so that every profiled launch belongs to a full forward and the per-kernel averages "
                         "are those of the `roofline` block)")
    ap.add_argument("--repeats", type=int, default=0,
                    help="the timed region of --steps steps (barrier + synchronize on both sides) is run this many times back to back; "
                         "`value` / `ms_per_step` are the MEDIAN region, the spread is reported next to it (SURVEY.md 8d: median + p5/p95).  "
                         "0 (default): as many regions as it takes for --min-timed-s seconds of timed GPU work (at least 5, at most 200), decided "
                         "from the first region -- so that the run is long enough for an outside utilisation sampler to see it")
    ap.add_argument("--min-timed-s", type=float, default=15.0, help="with --repeats 0: total duration of the timed regions to aim for")
    args = ap.parse_args(argv)
    argv = list(sys.argv[1:] if argv is None else argv)

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ      # torch.distributed.run, or our own spawn below
    if args.gpus > 1 and not launched:
        return self_launch(args, argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if os.environ.get("RD_BENCH_DRYRUN"):
        return dry_run(args, rank, world)

    import torch
    import torch.distributed as dist
    from rangedet_amd import dist as rdist, lib as rdlib, synth
    from rangedet_amd.pipeline import InterleavedPipelines

    gather = world > 1 or bool(os.environ.get("RD_BENCH_GATHER"))   # the env switch exercises the collective path on one GPU
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    cpus = rdist.bind_cpus(local, local_world)                      # per-rank CPU slice (reference: utils/cpu_affinity.py)
    dev = rdist.select_device(local, local_world)                   # HIP_VISIBLE_DEVICES-safe; raises instead of sharing a GPU
    if gather:
        rdist.init_process_group("nccl", dev)
    ranks_seen, rccl_version = [rank], None
    if gather:
        # every rank reports in through the communicator the data path uses (RCCL) with its rank and the identity of its GPU:
        # N distinct ranks on N distinct devices, or the run stops here
        ident = np.frombuffer(rdist.device_identity(dev), dtype=np.int32)
        me = torch.tensor([rank, local] + [int(v) for v in ident], device=dev, dtype=torch.int32)
        seen = torch.zeros((world, 6), device=dev, dtype=torch.int32)
        dist.all_gather_into_tensor(seen.view(-1), me)
        seen = seen.cpu().numpy()
        ranks_seen = [int(r) for r in seen[:, 0]]
        rdist.check_ranks(ranks_seen, [seen[r, 2:].tobytes() for r in range(world)] if world > 1 else [b"0"], world)
        try:
            rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:      # noqa: BLE001  (reporting only)
            rccl_version = None
    shard = rdist.FrameSharding(rank, world)
    dt = {"bf16": rdlib.RD_BF16, "f16": rdlib.RD_F16, "f32": rdlib.RD_F32}[args.dtype]

    global NPIX
    kitti = args.config == "kitti"
    Bf = args.batch
    if kitti:
        # KITTI range images (datasets/create_range_image_in_kitti.py:121,126): 64 x 2048, channels range, x, y, z, intensity
        from rangedet_amd.config.rangedet_veh_wo_aug_4_18e import KITTI_INPUT_CHANNELS
        Hk, Wk = 64, 2048
        NPIX = Hk * Wk
        params = synth.make_weights(seed=18, width=Wk, in_ch=KITTI_INPUT_CHANNELS, num_classes=2)
        pkw = dict(variant="kitti", feat_size=(Hk, Wk), pad_field=(Hk, Wk), pre_nms_top_n={'veh': 50000, 'ped': 5000})
    else:
        params = synth.make_weights(seed=18)
        pkw = {}
    multi = InterleavedPipelines(params, n=max(1, args.inflight), dtype=dt, wnms_cap=args.wnms_cap, batch=Bf,
                                 tie_order=args.tie_order, wnms_diag=rdlib.RD_WNMS_DIAG_NO_SKIP if args.wnms_no_skip else 0,
                                 graph=args.graph, **pkw)
    pipe = multi.pipes[0]   # (per-kernel profiling replay and the roofline figures use one pipeline on the default stream)
    # each rank owns its own frames (frame f -> rank f % world), resident in HBM before timing
    # (synthetic raw records through the device transform chain, rd_input_transform)
    if kitti:
        frames = [synth.make_batch(shard.frames_of_step(i, Bf), W=Wk, pad_W=Wk, H=Hk, lib=pipe.lib, alloc=pipe.alloc) for i in range(args.frames)]
        for f in frames:   # the Waymo-style synthetic record's channels in KITTI order: range, x, y, z, intensity
            f['input_data'] = f['input_data'][:, [0, 3, 4, 5, 1]].contiguous()
    else:
        frames = [synth.make_batch(shard.frames_of_step(i, Bf), lib=pipe.lib, alloc=pipe.alloc) for i in range(args.frames)]
    L = pipe.lib
    A = pipe.alloc
    h2d_mb = None
    if args.h2d:
        # PCIe-inclusive diagnostic: pinned host copies of every batch, one device input set per pipeline (a pipeline's previous batch
        # has been harvested before its inputs are overwritten), uploaded on the batch's launch stream in front of its forward
        if args.graph:
            raise SystemExit("bench.py: --h2d replays no graph (the upload is an eager copy in front of the forward)")
        host_in = [{k: v.cpu().pin_memory() for k, v in f.items()} for f in frames]
        dev_in = [{k: torch.empty_like(v) for k, v in frames[0].items()} for _ in multi.pipes]
        h2d_mb = sum(v.numel() * v.element_size() for v in frames[0].values()) / 1e6
    # RD_BENCH_GATHER_MODE=eager (A/B only): the collective enqueued WITH the batch, its communication stream waiting for the pack's event from
    # then on (DetectionGather.enqueue) -- the form of the first session of round 6, 7 % slower (profiles/r06o_gather_path_cost_bisect.txt)
    gather_eager = gather and os.environ.get("RD_BENCH_GATHER_MODE", "") == "eager"
    from collections import deque
    gather_pending = deque()        # (step, pipeline) of the batches whose pack is enqueued and whose collective is not: strictly in step order
    # one gather per pipeline and class (the two-class KITTI variant has two post-processors per pipeline)
    gathers = [[rdist.DetectionGather(p.bposts[c], shard, A, L) for c in p.class_names] for p in multi.pipes] if gather else None
    comm_streams = [torch.cuda.Stream(device=dev) for _ in multi.pipes] if gather else None   # one communication stream per pipeline
    # every step's results go to the host like the reference's loop materialises every frame (tools/test.py:151-153):
    # per pipeline, pinned host buffers for the (B, 200, 8) boxes, the keep counts and the candidate counts, filled by async
    # copies on the batch's post-processing stream
    classes = pipe.class_names                      # one class (waymo config) or vehicle + pedestrian (kitti)
    host = [{c: dict(d8=torch.empty((Bf, rdist.MAX_DET, 8), dtype=torch.float32).pin_memory(),
                     nkeep=torch.empty((Bf,), dtype=torch.int32).pin_memory(),
                     count=torch.empty((Bf,), dtype=torch.int32).pin_memory(),
                     stage=A.alloc(Bf * rdist.MAX_DET * 32)) for c in classes} for _ in multi.pipes]
    for h in host:
        h.update(done=None, step=-1)
    max_cand = [0]
    done_events = []
    import hashlib
    all_steps_sha = hashlib.sha256()      # every harvested step's host results (boxes of the kept detections, keep counts), in step order
    step_digests = []

    def issue_gathers(upto=None):
        """Enqueue the collectives of the batches whose done event has fired, oldest first and never out of step order (every rank issues the
        same sequence); upto = a step whose collective must be out when this returns (its event is waited for)."""
        while gather_pending:
            st_, j_ = gather_pending[0]
            ev_ = host[j_]["done_ev"]
            if upto is not None and st_ <= upto:
                ev_.synchronize()
            elif not ev_.query():
                break
            gather_pending.popleft()
            for g_ in gathers[j_]:
                g_.gather(comm_streams[j_])
            host[j_]["gdone"] = torch.cuda.Event()
            host[j_]["gdone"].record(comm_streams[j_])

    def harvest(j):
        """Host side of a finished batch: wait for its copies, check the WNMS capacity (K <= cap is what makes the
        device result the complete one), keep the largest candidate count for the report."""
        h = host[j]
        if h["done"] is None:
            return None
        h["done"].synchronize()
        if gather and not gather_eager:
            issue_gathers(upto=h["step"])
        for c in classes:
            kmax = int(h[c]["count"].max())
            if kmax > multi.pipes[j].bposts[c].cap:
                raise RuntimeError("step %d: %d %s candidates above min_score exceed --wnms-cap %d" % (h["step"], kmax, c, multi.pipes[j].bposts[c].cap))
            max_cand[0] = max(max_cand[0], kmax)
            nk = h[c]["nkeep"].numpy()
            one = hashlib.sha256(nk.tobytes())
            d8 = h[c]["d8"].numpy()
            for b_ in range(Bf):
                one.update(d8[b_, :min(int(nk[b_]), rdist.MAX_DET)].tobytes())
            all_steps_sha.update(one.digest())
            step_digests.append((h["step"], c, one.hexdigest()[:8], [int(v) for v in nk], [int(v) for v in h[c]["count"].numpy()]))
        h["done"] = None
        return h

    def step(i):
        # one step = one batch of Bf frames through the whole path; successive steps alternate between the pipelines
        # (two batches in flight: the other batch's launches fill the tails / launch gaps of this one)
        j = i % len(multi.pipes)
        harvest(j)                                         # the batch that last used this pipeline (two steps ago)
        if args.h2d:
            with multi.stream_context(j):
                for k, v in host_in[i % len(frames)].items():
                    dev_in[j][k].copy_(v, non_blocking=True)
            j2, _ = multi.enqueue(dev_in[j])
        else:
            j2, _ = multi.enqueue(frames[i % len(frames)])
        assert j2 == j
        pj, h = multi.pipes[j], host[j]
        with torch.cuda.stream(pj._post_stream):
            for c in classes:
                bp, hc = pj.bposts[c], h[c]
                nd = min(rdist.MAX_DET, bp.cap)
                L.call("rd_copy_rows", A.ptr(bp.out8), bp.cap * 32, A.ptr(hc["stage"]), rdist.MAX_DET * 32, 0, nd * 32, Bf,
                       pj._post_stream.cuda_stream)          # first 200 rows of every frame -> one contiguous block
                hc["d8"].copy_(A.view_f32(hc["stage"], (Bf, rdist.MAX_DET, 8)), non_blocking=True)
                hc["nkeep"].copy_(A.view_i32(bp.nkeep, (Bf,)), non_blocking=True)
                hc["count"].copy_(A.view_i32(bp.count, (Bf,)), non_blocking=True)
            done_on = pj._post_stream
            if gather:
                # the ONE collective of the path.  Its PACK (two strided copies) goes behind this batch's post-processing on its
                # post-processing stream (the batch's own launch stream with two or more batches in flight) -- after the previous gather of
                # this pipeline, which read the same record buffer (an event that fired n steps ago unless a peer rank is that late).  The
                # COLLECTIVE runs on the pipeline's communication stream -- no launch stream carries a collective -- and is issued by the
                # host once this batch's done event has fired (issue_gathers below), in step order on every rank: a communication stream
                # that waits for the pack from now on parks a barrier packet in a hardware queue for the whole forward, and that alone costs
                # 7 % of the throughput (RD_BENCH_GATHER_MODE=eager, profiles/r06o_gather_path_cost_bisect.txt)
                if h.get("gdone") is not None and not gather_eager:
                    pj._post_stream.wait_event(h["gdone"])
                for g_ in gathers[j]:
                    if gather_eager:
                        done_on = g_.enqueue(pj._post_stream, comm_streams[j])
                    else:
                        g_.pack(pj._post_stream)
                if not gather_eager:
                    gather_pending.append((i, j))
            h["done"] = torch.cuda.Event(enable_timing=True)
            h["done"].record(done_on)
            h["done_ev"] = h["done"]           # (harvest() clears "done"; the pending collective of this batch keeps the event)
            h["step"] = i
            done_events.append(h["done"])      # (completion time of every step: the per-step percentiles of the report)
        if gather and not gather_eager:
            issue_gathers()                    # whatever has finished in the meantime (usually the batch before last)

    def barrier():
        if gather and not gather_eager:
            issue_gathers(upto=1 << 60)        # the collective of every enqueued batch is out (and, below, complete) inside the timed region
        torch.cuda.synchronize(dev)
        if gather:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if args.graph:
        # set-up, not a step (nothing is harvested): every pipeline meets every input set once -- that enqueue runs eagerly and captures the
        # graph (a device-wide synchronisation each) -- so that every warm-up and timed step below is a replay
        for j_, p_ in enumerate(multi.pipes):
            for f_ in frames:
                with multi.stream_context(j_):
                    p_.enqueue(f_)
        torch.cuda.synchronize(dev)
    for i in range(args.warmup):
        step(i)
    barrier()
    if gather:   # RCCL prints a version banner through C stdio when the communicator comes up: get it out of the way now
        import ctypes
        ctypes.CDLL(None).fflush(None)
    # The timed region = EXACTLY --steps steps between two barriers (+ synchronize on both sides), max over ranks.  It is run
    # --repeats times back to back (step numbering continues, so the pipelines keep alternating); the headline is the MEDIAN
    # region, and the completion event of every step gives the per-step distribution (interval between consecutive completions
    # inside a region: with two batches in flight that is the steady-state time per step).
    repeats = max(1, args.repeats) if args.repeats > 0 else 5
    region_s, step_ms = [], []
    nstep = args.warmup
    r = 0
    while r < repeats:
        r += 1
        del done_events[:]
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(nstep)
            nstep += 1
        barrier()
        for j in sorted(range(len(multi.pipes)), key=lambda j_: host[j_].get("step", -1)):     # in step order: the all-steps digest does not
            harvest(j)                                                                             # depend on --inflight
        region_s.append(time.perf_counter() - t0)
        if args.repeats <= 0 and r == 1:      # auto: the number of regions from the first one (the same decision on every rank)
            first = torch.tensor([region_s[0]], device=dev, dtype=torch.float64)
            if gather:
                dist.all_reduce(first, op=dist.ReduceOp.MAX)
            repeats = int(min(200, max(5, np.ceil(args.min_timed_s / max(float(first.item()), 1e-4)))))
        # steady-state time per step from the completion events: with n batches in flight on n pipelines the completions come in
        # bursts of n (the streams share the GPU and finish together), so the interval is taken over a window of n steps and divided
        # by n -- (steps - n) samples per region
        nw = len(multi.pipes)
        step_ms += [done_events[i].elapsed_time(done_events[i + nw]) / nw for i in range(len(done_events) - nw)]
    region_by_rank = None
    if gather:
        t = torch.tensor(region_s, device=dev, dtype=torch.float64)
        allr = torch.empty(world * len(region_s), device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allr, t)                  # every rank's own clock around every region: stragglers show as a spread
        per = allr.view(world, -1).cpu().numpy() * 1e3
        region_by_rank = {"median_ms_per_rank": [round(float(np.median(r_)), 3) for r_ in per],
                          "min_ms": round(float(per.min()), 3), "max_ms": round(float(per.max()), 3)}
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        region_s = [float(v) for v in t.cpu()]
    elapsed = float(np.median(region_s))
    total_steps = args.warmup + args.steps * repeats
    last = (total_steps - 1) % len(multi.pipes)
    res = multi.pipes[last].collect()[0]
    gathered_frames = gather_matches_local = None
    # digest of the last step's device results (every frame's kept rows, counts): the same data with and without the collective
    import hashlib
    lastf = multi.pipes[last].bposts[classes[0]].collect_all()
    results_sha256 = hashlib.sha256(b"".join(f["wnms_rows"].tobytes() + f["keep_inds"].tobytes() for f in lastf)).hexdigest()[:16]
    if gather:
        # what came out of the collective against what this rank put in: its own B frames' records, bit for bit
        got = gathers[last][0].unpack(total_steps - 1)
        gathered_frames = len(got)
        gather_matches_local = True
        for b_, f_ in enumerate(shard.frames_of_step(total_steps - 1, Bf)):
            rows_, M_ = got[f_]
            want_ = lastf[b_]["wnms_rows"][:rdist.MAX_DET]
            gather_matches_local = gather_matches_local and M_ == len(lastf[b_]["wnms_rows"]) and np.array_equal(rows_.view(np.uint32), want_.view(np.uint32))

    # ---- per-kernel timing with HIP events on the launch stream, over a replay of the same steps ------------------
    roof = meta_info = prof = backbone_info = block_info = None
    if rank == 0:
        # the forwards of the timed steps back to back on ONE stream with nothing else on the GPU, then the post-processing of those
        # batches the same way: every kernel runs alone at a steady clock, which is what a `rocprofv3 --kernel-trace --stats` pass
        # of `bench.py --inflight 1` reports (profiles/).  In the timed region two batches overlap on two launch streams plus the
        # side stream: each launch's start-to-end time stretches there while the step gets SHORTER than the serial sum.
        L.call("rd_prof_reset")
        L.call("rd_prof_enable", 1)
        nprof = min(args.steps, 20)
        with multi.stream_context(0):
            for i in range(nprof):
                pipe.exe.forward(frames[i % len(frames)])
            torch.cuda.synchronize(dev)
            for i in range(nprof):
                for c in classes:
                    pipe.bposts[c].enqueue_nms()
        torch.cuda.synchronize(dev)
        prof = L.prof()
        L.call("rd_prof_enable", 0)
        # the event pair around every launch is itself a pair of queue packets between the kernels (~10 us per launch on this
        # stack: 120 us by events against 106 - 108 us per launch in the rocprofv3 trace and in the un-instrumented replay below), so
        # the per-kind event sums only give each kind's SHARE; the forward's wall time is taken from the same replay without them
        with multi.stream_context(0):
            pipe.exe.forward(frames[0])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(nprof):
                pipe.exe.forward(frames[i % len(frames)])
            e1.record()
        torch.cuda.synchronize(dev)
        fwd_ms = e0.elapsed_time(e1) / nprof
        fwd_kinds = ("conv", "conv3", "block", "meta", "head_out", "sort", "decode", "layout")
        ev_fwd_ms = sum(prof[k][0] for k in fwd_kinds) / nprof
        scale = fwd_ms / ev_fwd_ms if ev_fwd_ms else 1.0
        prof = {k: ((v[0] * scale, v[1]) if k in fwd_kinds else v) for k, v in prof.items()}
        bf = dt in rdlib.H16
        # dominant kernel: bf16 = the persistent 3x3 stride-1 kernel (own profiling kind); f32 = the generic tap kernel
        fl, nlaunch = conv_flops(pipe.plan, only_conv3=bf)
        ms, cnt = prof["conv3" if bf else "conv"]
        avg_ms = ms / max(cnt, 1)
        achieved = (fl * Bf / nlaunch) / (avg_ms * 1e-3) / 1e12 if cnt else 0.0   # a launch covers the Bf frames of the batch
        fl_all, n_all = conv_flops(pipe.plan)
        ms_all = prof["conv"][0] + prof["conv3"][0] + prof["block"][0]
        # the fused BasicBlock launches (block64_stream_kernel, csrc/k_block.h): algorithmic FLOPs / bytes of the two convs each replaces
        fl_b, n_b = conv_flops(pipe.plan, only_conv3="block")
        ms_b, cnt_b = prof["block"]
        block_info = None
        if n_b:
            avg_b = ms_b / max(cnt_b, 1)
            by_b = conv_bytes(pipe.plan, 2 if bf else 4, only_conv3="block") * Bf / n_b
            block_info = {"kernel": "block64_stream_kernel (conv1 + BN + ReLU + conv2 + BN + shortcut + ReLU of a 64-channel BasicBlock, the "
                                    "intermediate tensor in LDS)", "launches_per_step": n_b, "avg_launch_ms": avg_b,
                          "gflop_per_launch": fl_b * Bf / n_b / 1e9, "tflops_algorithmic": (fl_b * Bf / n_b) / (avg_b * 1e-3) / 1e12 if cnt_b else 0.0,
                          "algorithmic_bytes_per_launch_unfused_model": by_b,
                          "traffic": measured_traffic("block64_stream_kernel", Bf),     # HBM bytes per launch by the PMC passes (null: kernels changed since)
                          "note": "FLOPs = those of the two 3x3 convs (+ 1x1 shortcut) the launch replaces; the kernel executes 1.25x that "
                                  "(conv1 on the 10 x 34 halo of every 8 x 32 tile)"}
        roof = {"kernel": "conv3x3_stream_kernel (persistent 3x3 implicit-GEMM conv / transposed-conv phase + BN + ReLU + residual)" if bf
                else "conv_taps_kernel (implicit-GEMM conv/deconv + BN + ReLU + residual)", "bound": "mfma",
                "achieved": achieved, "peak": PEAK_BF16_TFLOPS if bf else 157.3, "unit": "TFLOP/s",
                "frac": achieved / (PEAK_BF16_TFLOPS if bf else 157.3),
                "traffic": measured_traffic("conv3x3_stream_kernel", Bf) if bf else None,
                "traffic_note": "HBM bytes per launch, (2*FETCH_SIZE + WRITE_SIZE) KiB from separate rocprofv3 --pmc passes of "
                                "this command (newest profiles/r*_pmc_traffic.json whose csrc_sha16 equals the running sources' "
                                "hash, tools/profile_round.sh; null when the kernels changed since the last PMC pass)",
                "algorithmic_bytes_per_launch": conv_bytes(pipe.plan, 2 if bf else 4, only_conv3=bf) * Bf / nlaunch,
                "launches_per_step": nlaunch, "avg_launch_ms": avg_ms, "gflop_per_launch": fl * Bf / nlaunch / 1e9,
                "serial_ms_per_step": avg_ms * nlaunch, "ms_per_step": elapsed / args.steps * 1e3,
                # the same launches priced by the TIMED region: the step's wall time shared out over all kernels in proportion to
                # their stand-alone durations, so that the shares of a step add up to ms_per_step
                "in_step_launch_ms": (elapsed / args.steps * 1e3) * (ms / max(sum(v[0] for v in prof.values()), 1e-9)) / max(nlaunch, 1),
                "achieved_in_step": (fl * Bf / nlaunch) / ((elapsed / args.steps) * (ms / max(sum(v[0] for v in prof.values()), 1e-9)) / max(nlaunch, 1)) / 1e12 if cnt else 0.0,
                "serial_forward_ms": fwd_ms, "event_overhead_factor": 1.0 / scale,
                "timing_note": "avg_launch_ms / achieved / frac: HIP events around every launch of the kernel in a serial replay of the "
                               "timed steps on one stream with nothing else on the GPU give each kernel kind's share of the forward; the "
                               "forward's wall time comes from the same replay without the per-launch events (serial_forward_ms; the event "
                               "packets between the kernels cost event_overhead_factor).  The kernel alone: agrees with the rocprofv3 "
                               "--inflight 1 summary under profiles/.  serial_ms_per_step = avg_launch_ms x launches_per_step may exceed "
                               "ms_per_step: the timed region overlaps the batches in flight (--inflight, default 3) on their launch streams, which fills the tail rounds and "
                               "launch gaps of the serial order.  in_step_launch_ms / achieved_in_step: the step's measured wall time shared "
                               "out over all kernels in proportion to their stand-alone durations (the shares add up to ms_per_step)",
                "share_of_conv_flops": fl / fl_all,
                "all_conv_family": {"launches_per_step": n_all, "gflop_per_frame": fl_all / 1e9,
                                    "tflops": fl_all * Bf * nprof / (ms_all * 1e-3) / 1e12 if ms_all else 0.0}}
        mms, mcnt = prof["meta"]
        esz = 2 if dt in rdlib.H16 else 4
        mbytes = Bf * NPIX * ((64 + 64) * esz + 3 * 4)  # compulsory: data in + out, coords fp32 (262 B/px bf16)
        meta_info = {"kernel": "meta_kernel (fused Meta-Kernel unit)", "bound": "hbm",
                     "achieved": mbytes / (mms / max(mcnt, 1) * 1e-3) / 1e9 if mcnt else 0.0, "peak": PEAK_HBM_GBPS,
                     "unit": "GB/s", "avg_launch_ms": mms / max(mcnt, 1), "bytes_per_launch": mbytes,
                     "tflops": Bf * META_FLOP_PER_PX * NPIX / (mms / max(mcnt, 1) * 1e-3) / 1e12 if mcnt else 0.0}
        meta_info["frac"] = meta_info["achieved"] / PEAK_HBM_GBPS
        meta_info["intensity_flop_per_byte"] = META_FLOP_PER_PX * NPIX * Bf / mbytes
        meta_info["traffic"] = measured_traffic("meta_kernel", Bf) if dt == rdlib.RD_BF16 else None
        backbone_info = backbone_forward_roofline(pipe, frames[0], Bf, esz, reps=args.backbone_reps) if args.backbone_reps > 0 else None
        # headline figure of the Meta-Kernel = the kernel with nothing else resident; in the pipeline the previous batch's NMS
        # kernels (side stream) co-run with it and its launch-to-end time is longer -- both are reported
        alone = backbone_info.pop("meta_kernel_alone") if backbone_info else None
        if alone:
          meta_info.update({"in_pipeline_avg_launch_ms": meta_info["avg_launch_ms"], "in_pipeline_gbps": meta_info["achieved"],
                            "avg_launch_ms": alone["avg_launch_ms"], "achieved": alone["gbps"], "frac": alone["frac_hbm_peak"],
                            "tflops": Bf * META_FLOP_PER_PX * NPIX / (alone["avg_launch_ms"] * 1e-3) / 1e12,
                            "frac_mfma_peak": Bf * META_FLOP_PER_PX * NPIX / (alone["avg_launch_ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                            "binding_roof": "mfma (433 FLOP/B is above the ridge of 312 FLOP/B); in practice the latency of its dependent "
                                            "MFMA -> vector -> MFMA chains at two waves per SIMD (DESIGN.md section 6.3)",
                            "note": "achieved / avg_launch_ms: serial replay of the kernel alone (HIP events); in_pipeline_*: the "
                                  "same launch inside the timed pipeline, where the previous batch's NMS kernels share the GPU"})

    if rank == 0:
        out = {
            "metric": "range-image frames/sec (64x2650, 8ch) at 1/2/4/8 GPU; Meta-Kernel HBM GB/s vs peak",
            "value": args.steps * world * Bf / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            # spread: `value` / `ms_per_step` are the median of `repeats` timed regions of `steps` steps each;
            # ms_per_step_p5/p50/p95 come from the completion events of the individual steps (interval between the completions of
            # steps i and i + n over n = batches in flight, (steps - n) x repeats samples; own-rank events -- rank 0 for N > 1)
            "repeats": repeats, "value_min": args.steps * world * Bf / max(region_s), "value_max": args.steps * world * Bf / min(region_s),
            "region_ms": [round(v * 1e3, 3) for v in region_s], "region_ms_by_rank": region_by_rank,
            "timed_gpu_s": round(float(sum(region_s)), 3),      # all timed regions together: what an outside GPU-utilisation sampler can see
            "ms_per_step_p5": float(np.percentile(step_ms, 5)) if step_ms else None,
            "ms_per_step_p50": float(np.percentile(step_ms, 50)) if step_ms else None,
            "ms_per_step_p95": float(np.percentile(step_ms, 95)) if step_ms else None,
            "value_p50": (world * Bf / (float(np.percentile(step_ms, 50)) * 1e-3)) if step_ms else None,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("KITTI-shaped (BASELINE configs[4]): DLA backbone + Meta-Kernel + vehicle and pedestrian heads + per-class "
                                    "top-k (50000 / 5000) + 3D decode + weighted NMS per class on 64x2048 x 5ch synthetic range images, "
                                    "%d frames per step per GPU, random-init weights (seed 18)" % Bf) if kitti else
                                   "rangedet_veh_wo_aug_4_18e: DLA backbone + Meta-Kernel + heads + top-50000 + 3D decode "
                                   "+ weighted NMS on 64x2650 (pad 2656) x 8ch synthetic range images, %d frames per step per GPU, " % Bf + ""
                                   "random-init weights (seed 18)", "frames_per_step": world * Bf, "frames_per_gpu_per_step": Bf, "batches_in_flight_per_gpu": len(multi.pipes), "parallelism": "frame-parallel dp%d" % world,
                       "wnms_candidates": int(res["num_candidates"]), "wnms_kept": int(len(res["keep_inds"])),
                       "per_class": {c: {"candidates": int(r["num_candidates"]), "kept": int(len(r["keep_inds"]))}
                                     for c, r in res.get("per_class", {}).items()} or None,
                       "hip_graph": {"replays": int(sum(p_.graph_replays for p_ in multi.pipes)), "graphs": int(sum(len(p_._graphs) for p_ in multi.pipes))} if args.graph else None,
                       "wnms_cap": int(pipe.bpost.cap), "wnms_tie_order": args.tie_order, "max_candidates_seen": int(max_cand[0]),
                       "inputs": ("uploaded every step from pinned host memory on the batch's launch stream: %.1f MB per step (PCIe-inclusive diagnostic)" % h2d_mb
                                  if args.h2d else "resident in HBM before the timed region"),
                       "results_to_host": "every step: (B,200,8) boxes + counts, async D2H on the post-processing stream into pinned memory, K <= cap checked",
                       "gathered_frames_last_step": gathered_frames, "gather_matches_local": gather_matches_local,
                       "results_sha256_last_step": results_sha256, "results_sha256_all_steps": all_steps_sha.hexdigest()[:16],
                       "step_digests": step_digests if os.environ.get("RD_BENCH_STEP_DIGESTS") else None, "ranks_seen": ranks_seen, "rccl_version": rccl_version,
                       "cpus_per_rank": len(cpus) if cpus else None,
                       "launcher": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else
                                   ("self (mp.spawn)" if world > 1 else "single process")},
            "roofline": roof, "block_kernel": block_info, "meta_kernel": meta_info, "meta_dla_forward": backbone_info,
            # the whole path against both roofs: algorithmic conv-family bytes / flops of a frame + the
            # Meta-Kernel's, over the measured wall time per frame (everything included: NMS, launches, side stream)
            "path_roofline": (lambda sec, gb, gf: {"ms_per_frame": sec * 1e3, "algorithmic_gb_per_frame": gb,
                                                   "hbm_gbps": gb / sec, "frac_hbm_peak": gb / sec / PEAK_HBM_GBPS,
                                                   "gflop_per_frame": gf, "tflops": gf / sec / 1e3,
                                                   "frac_mfma_peak": gf / sec / 1e3 / PEAK_BF16_TFLOPS})(
                elapsed / (args.steps * world * Bf) * world,
                (conv_bytes(pipe.plan, 2 if dt in rdlib.H16 else 4) + NPIX * 262) / 1e9,
                (conv_flops(pipe.plan)[0] + META_FLOP_PER_PX * NPIX) / 1e9),
            "kernel_ms_per_frame": {k: v[0] / max(1, min(args.steps, 20)) / Bf for k, v in prof.items()},
        }
        if not args.no_cpu_baseline and world == 1 and not kitti:
            from oracle import input_ref   # (the checker's numpy transform: only this leg may touch oracle/)
            out["cpu_baseline"] = cpu_baseline(params, [input_ref.make_frame(i) for i in range(3)])
        line = json.dumps(out)
    if gather:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner through C stdio, which would otherwise be flushed at exit, AFTER this line
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        if world > 1:
            time.sleep(1.0)   # let the other ranks' exit-time output drain first: this line should be the last one
        print(line, flush=True)


if __name__ == "__main__":
    main()
