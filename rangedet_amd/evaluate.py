"""The reference's evaluation loop (tools/test.py:84-238) on the HIP path, without MXNet: roidb records -> device input
transform -> forward + post-processing -> `output_dict` / `annotation_dict` -> the pickle tools/create_prediction_bin_3d.py
(rangedet_amd.export) reads.

    python -m rangedet_amd.evaluate --roidb 'data/validation/*.roidb' --prefix experiments/<cfg>/checkpoint --epoch 18 \
        [--out experiments/<cfg>/checkpoint_output_dict_18e.pkl] [--bin-dir <dir> --config-name <cfg>] [--batch 8] [--gpus 8]
    python -m rangedet_amd.evaluate --synthetic 16 --random-weights --out /tmp/out.pkl         (self-contained dry run)

A roidb record is the dict datasets/create_range_image_roidb.py writes: `pc_url` (npz with the arrays `range_image`,
`pc_vehicle_frame`, `inclination`, `azimuth`, ...: create_range_image_roidb.py:119-124,164; read by LoadRecord,
rangedet/core/input.py:23-38), `gt_bbox_imu`, `gt_class`.  Frames are batched; each frame's
result is what tools/test.py:200-232 computes for it: `det_xyzlwhyaws[TYPE_VEHICLE]` (M,8) + `meta_info`.
"""
import argparse
import glob
import pickle as pkl

import numpy as np

mapping = {'veh': 'TYPE_VEHICLE', 'ped': 'TYPE_PEDESTRIAN', 'cyc': 'TYPE_CYCLIST'}   # tools/test.py:170


def load_record(rec):
    """LoadRecord (rangedet/core/input.py:14-42) for one roidb entry: the raw arrays the device transform needs."""
    if 'range_image' in rec:                                   # already loaded (synthetic records)
        return rec
    with np.load(rec['pc_url']) as z:                          # same keys and float32 casts as LoadRecord.apply (:29-36)
        return dict(range_image=z['range_image'].astype(np.float32), pc_vehicle_frame=z['pc_vehicle_frame'].astype(np.float32),
                    inclination=z['inclination'].astype(np.float32))


class RecordPrefetcher:
    """Batches of loaded records, read `depth` batches ahead of the consumer by `threads` worker threads (np.load of a frame's
    npz is ~14 MB of zip inflate, which releases the GIL).  The reference overlaps loading with the forward through the worker
    threads of its Loader (tools/test.py:120-137, rangedet/core/loader.py -- out of scope as a component); without this the
    enqueue thread would load 8 frames between two enqueues, and the loop would be I/O bound far below the GPU's rate.
    Order is preserved; a loading error surfaces at the batch that needed the record."""

    def __init__(self, roidb, chunks, threads=4, depth=2):
        from concurrent.futures import ThreadPoolExecutor
        self.roidb, self.chunks, self.depth = roidb, chunks, max(1, depth)
        self.pool = ThreadPoolExecutor(max_workers=max(1, threads)) if threads > 0 else None
        self.futs, self.next = [], 0

    def _submit(self):
        while self.next < len(self.chunks) and len(self.futs) < self.depth:
            chunk = self.chunks[self.next]
            self.futs.append([self.pool.submit(load_record, self.roidb[i]) for i in chunk])
            self.next += 1

    def __iter__(self):
        try:
            for k, chunk in enumerate(self.chunks):
                if self.pool is None:
                    yield chunk, [load_record(self.roidb[i]) for i in chunk]
                    continue
                self._submit()
                futs = self.futs.pop(0)
                self._submit()                                  # keep `depth` batches in flight while this one is consumed
                yield chunk, [f.result() for f in futs]
        finally:
            if self.pool is not None:
                self.pool.shutdown(wait=False, cancel_futures=True)


def meta_info(rec, rid):
    url = rec.get('pc_url')
    if not url:
        return {'name': 'synthetic', 'timestamp_micros': int(rid)}
    name = url.split('/')[-2].replace('segment-', '').replace('_with_camera_labels', '')   # tools/test.py:226-228
    return {'name': name, 'timestamp_micros': int(url.split('/')[-1][:-4])}


def run(roidb, params, batch=8, variant='veh', wnms=True, progress=None, pre_nms_top_n=50000, shard=None, inflight=3,
        wnms_cap=None, loader_threads=4):
    """-> (annotation_dict, output_dict) exactly as tools/test.py:166-233 builds them (frames without detections are absent).

    shard (rangedet_amd.dist.FrameSharding): this process handles the records shard.mine(len(roidb)) -- the reference runs one
    DetModule per GPU off a shared queue (tools/test.py:143-161); here one process per GPU owns every world-th record and the
    per-rank dictionaries are merged by merge_across_ranks.  Batches overlap: `inflight` pipelines on their own streams
    (pipeline.InterleavedPipelines), batch i+1 is enqueued before batch i's results are read back.  A batch in which a frame has
    more candidates above min_score than the weighted NMS was sized for is re-run as a whole with a capacity that fits (the
    reference has no such limit, nms.h:452-577), never truncated."""
    from . import lib as rdlib
    from .input_transform import DeviceInputTransform
    from .pipeline import InterleavedPipelines, RangeDetPipeline
    mine = list(range(len(roidb))) if shard is None else shard.mine(len(roidb))
    output_dict, annotation_dict = {}, {}
    if not mine:
        return annotation_dict, output_dict
    H, W = np.asarray(load_record(roidb[mine[0]])['range_image']).shape[:2]
    Wp = -(-W // 32) * 32
    from .config import rangedet_veh_wo_aug_4_18e as cfgmod
    # pre_nms_top_n: an int (the first class) or {class: k} (two-class variant); capacities are sized for the largest class
    topn = dict(pre_nms_top_n) if isinstance(pre_nms_top_n, dict) else {cfgmod.variant_classes(variant)[0]: int(pre_nms_top_n)}
    cap = min(wnms_cap or 8192, max(topn.values()))
    kw = dict(batch=batch, feat_size=(H, W), pad_field=(H, Wp), variant=variant, wnms=wnms, pre_nms_top_n=pre_nms_top_n)
    multi = InterleavedPipelines(params, n=max(1, inflight), wnms_cap=cap, **kw)
    pipe0 = multi.pipes[0]
    to_inputs = DeviceInputTransform(pad_hw=(H, Wp), lib=pipe0.lib, alloc=pipe0.alloc)
    cls = mapping[variant if variant in mapping else 'veh']
    big = {}                                                   # (capacity, batch) -> pipeline sized for the worst case, for overflowing batches

    def rerun(j, inputs):
        """The WHOLE batch again through a pipeline whose weighted NMS is sized for the worst case (every pre-NMS candidate above
        min_score): one forward + one batched NMS instead of `batch` single-frame runs.  Built on first use and kept; its workspace
        is 3 K^2 / 8 bytes per frame (0.94 GB at K = 50 000, so 7.5 GB for a batch of 8 -- small against
        288 GB, but it is allocated in the middle of a run with `inflight` pipelines resident: when that allocation fails (a smaller or
        shared GPU) the batch is re-run frame by frame through a single-frame pipeline instead (0.94 GB).  Enqueued on pipeline j's
        launch stream, i.e. ordered behind the batch whose overflow it repairs."""
        K = min(max(topn.values()), rdlib.RD_WNMS_MAX_K)
        with multi.stream_context(j):
            if (K, batch) not in big and (K, 0) not in big:
                try:
                    big[(K, batch)] = RangeDetPipeline(params, wnms_cap=K, **kw)
                except (MemoryError, RuntimeError) as e:       # torch.cuda.OutOfMemoryError is a RuntimeError
                    if "out of memory" not in str(e).lower() and not isinstance(e, MemoryError):
                        raise
                    big[(K, 0)] = None                         # remember: the whole-batch form does not fit here
            if (K, batch) in big:
                big[(K, batch)].enqueue(inputs)
                return big[(K, batch)].collect()
            if (K, 1) not in big:
                big[(K, 1)] = RangeDetPipeline(params, wnms_cap=K, **dict(kw, batch=1))
            frames = []
            for b in range(batch):
                big[(K, 1)].enqueue({k_: v_[b:b + 1] for k_, v_ in inputs.items()})
                frames += big[(K, 1)].collect()
            return frames

    def finish(j, chunk, recs, inputs):
        # one wait (for this pipeline's own post-processing event) and one host copy per batch
        try:
            frames = multi.pipes[j].collect()
        except rdlib.RangeDetError as e:
            if e.code != rdlib.RD_EWORKSPACE or not wnms:
                raise
            frames = rerun(j, inputs)                          # a frame above the WNMS capacity: this batch again, sized for the worst case
        for b, i in enumerate(chunk):
            rec, fr = roidb[i], frames[b]
            rid = rec.get('rec_id', i)
            per = fr.get('per_class') or {pipe0.class_names[0]: fr}
            det = {mapping[c]: r['det_xyzlwhyaws'] for c, r in per.items() if r['det_xyzlwhyaws'].shape[0]}
            if not det:
                continue                                       # tools/test.py:204-205, 222-223
            output_dict[rid] = {'det_xyzlwhyaws': det, 'meta_info': meta_info(rec, rid)}
            annotation_dict[rid] = rec.get('gt_bbox_imu')

    pending = []                                               # (pipeline index, record indices, records, inputs kept alive)
    done = 0
    chunks = [mine[i0:i0 + batch] for i0 in range(0, len(mine), batch)]
    for chunk, recs in RecordPrefetcher(roidb, chunks, threads=loader_threads, depth=max(2, inflight)):
        padded = recs + [recs[-1]] * (batch - len(recs))       # the last batch is padded with its last frame
        if len(pending) == len(multi.pipes):                   # the pipeline about to be reused must be read back first
            finish(*pending.pop(0))
        with multi.stream_context(multi._i % len(multi.pipes)):
            inputs = to_inputs(padded)                         # the transform runs on the batch's own launch stream
        j, _ = multi.enqueue(inputs)
        pending.append((j, chunk, recs, inputs))
        done += len(chunk)
        if progress:
            progress(done, len(mine))
    while pending:
        finish(*pending.pop(0))
    return annotation_dict, output_dict


def merge_across_ranks(annotation_dict, output_dict, force=False):
    """All ranks' (annotation_dict, output_dict) merged on every rank (keys are global record ids, disjoint across ranks).
    force: run the collective with one rank too (RD_EVAL_GATHER)."""
    import torch.distributed as dist
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return annotation_dict, output_dict
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, (annotation_dict, output_dict))
    ann, out = {}, {}
    for a, o in parts:
        if set(a) & set(ann) or set(o) & set(out):
            raise RuntimeError("two ranks produced the same record id")
        ann.update(a)
        out.update(o)
    return dict(sorted(ann.items())), dict(sorted(out.items()))


def _rank_main(rank, world, port, argv):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    main(argv, _spawned=True)


def main(argv=None, _spawned=False):
    ap = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    ap.add_argument('--gpus', type=int, default=1, help="one process per GPU; records are sharded rank = index % gpus "
                                                        "(under torch.distributed.run the launcher's world size is used)")
    ap.add_argument('--loader-threads', type=int, default=4, help="threads reading record npz files ahead of the GPU (0: load on the enqueue thread)")
    ap.add_argument('--roidb', help="glob of .roidb pickles (lists of records)")
    ap.add_argument('--synthetic', type=int, default=0, help="use N synthetic records instead of --roidb")
    ap.add_argument('--prefix'), ap.add_argument('--epoch', type=int)
    ap.add_argument('--random-weights', action='store_true')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--out', required=True, help="pickle path (annotation_dict, then output_dict: tools/test.py:235-237)")
    ap.add_argument('--bin-dir'), ap.add_argument('--config-name', default='rangedet_veh_wo_aug_4_18e')
    ap.add_argument('--nms3d', action='store_true', help="RpnParam.wnms = False: contrib.NMS3D instead of the weighted NMS")
    a = ap.parse_args(argv)
    import os
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ          # torch.distributed.run or our own spawn
    if a.gpus > 1 and not launched:
        import socket
        import torch.multiprocessing as mp
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        mp.spawn(_rank_main, args=(a.gpus, port, list(argv) if argv is not None else __import__("sys").argv[1:]), nprocs=a.gpus)
        return
    shard = None
    # RD_EVAL_GATHER=1 (like bench.py's RD_BENCH_GATHER): take the multi-rank branch with ONE rank -- RCCL communicator on the GPU,
    # sharding, the merge of the per-rank dictionaries through the collective -- so that the path runs on a one-GPU box
    force = bool(os.environ.get("RD_EVAL_GATHER"))
    if force and not launched:
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        launched = True
    if launched and (int(os.environ["WORLD_SIZE"]) > 1 or force):
        import torch
        from . import dist as rdist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ["WORLD_SIZE"]))
        rdist.bind_cpus(local, local_world)                     # per-rank CPU slice (reference: utils/cpu_affinity.py)
        dev = rdist.select_device(local, local_world)           # raises instead of letting two ranks share a GPU
        rdist.init_process_group("nccl", dev)
        shard = rdist.FrameSharding()
    from . import synth
    if a.synthetic:
        roidb = [dict(synth.raw_record(i), rec_id=i) for i in range(a.synthetic)]
    else:
        roidb = []
        for s in sorted(glob.glob(a.roidb)):
            roidb += pkl.load(open(s, 'rb'), encoding='latin1')
        for i, r in enumerate(roidb):
            r['rec_id'] = i                                     # tools/test.py:113-114
    if a.random_weights:
        params = synth.make_weights(seed=18)
    else:
        from .load_model import load_params
        params = load_params(a.prefix, a.epoch)
    rank = shard.rank if shard else 0
    ann, out = run(roidb, params, batch=a.batch, wnms=not a.nms3d, shard=shard, loader_threads=a.loader_threads,
                   progress=(lambda d, n: print('%d of %d records' % (d, n), flush=True)) if rank == 0 else None)
    ann, out = merge_across_ranks(ann, out, force=force)
    if rank == 0:
        with open(a.out, 'wb') as fw:
            pkl.dump(ann, fw)
            pkl.dump(out, fw)
        print('%d frames with detections of %d -> %s' % (len(out), len(roidb), a.out))
        if shard is not None:
            import torch.distributed as dist
            print('merged over %d rank(s) through the %s communicator' % (dist.get_world_size(), dist.get_backend()))
        if a.bin_dir:
            from . import export
            export.main(a.out, a.config_name, a.bin_dir)
    if shard is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
