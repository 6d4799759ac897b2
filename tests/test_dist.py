"""N > 1 path on CPU: world-size-2 gloo run of the frame sharding + padded all_gather logic bench.py uses
(no GPU, no kernels: the per-frame results are stand-ins; what is tested is the distribution contract)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

MAX_DET = 200


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nframes, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = [f for f in range(nframes) if f % world == rank]          # frame f -> rank f % world
    gathered = {}
    for step, f in enumerate(mine):
        rng = np.random.default_rng(f)
        M = int(rng.integers(0, MAX_DET))
        rows = np.zeros((MAX_DET, 12), np.float32)
        rows[:M] = rng.standard_normal((M, 12))
        payload = torch.from_numpy(np.concatenate([rows.reshape(-1), [float(M)], [float(f)]]).astype(np.float32))
        bufs = [torch.zeros_like(payload) for _ in range(world)]
        dist.all_gather(bufs, payload)                                # the one collective of the path
        for b in bufs:
            gathered[int(b[-1])] = (int(b[-2]), b[:-2].numpy().reshape(MAX_DET, 12).copy())
    t = torch.tensor([float(len(mine))])
    dist.all_reduce(t)                                                # max/sum-over-ranks bookkeeping like bench.py
    if rank == 0:
        out.put((t.item(), {k: (v[0], float(np.abs(v[1]).sum())) for k, v in gathered.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharding_and_gather_gloo():
    world, nframes = 2, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nframes, q)) for r in range(world)]
    for p in procs:
        p.start()
    total, got = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert total == nframes and sorted(got) == list(range(nframes))   # every frame processed exactly once, all visible on rank 0
    for f, (M, checksum) in got.items():
        rng = np.random.default_rng(f)
        M_ref = int(rng.integers(0, MAX_DET))
        rows = rng.standard_normal((M_ref, 12)).astype(np.float32)
        assert M == M_ref and abs(checksum - float(np.abs(rows).sum())) < 1e-2
