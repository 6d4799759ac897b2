"""Host-side enqueue cost of one 8-frame step, alone and with N processes enqueueing at once (dev tool).

The 8-GPU run is one Python process per GPU on ONE host: every process issues ~95 C-ABI launches + the batch's result copies
per step.  The path scales only if that host-side work stays well below the step's GPU time when 8 interpreters run side
by side.  This probe measures exactly that part -- the time a process needs to ENQUEUE a step (no synchronisation inside the
timed region) -- for N concurrent processes; on a one-GPU box they all enqueue to cuda:0 (the GPU then serialises their
work, which does not matter: the enqueue calls return as soon as the launches are queued).

    python tools/enqueue_time.py [--procs 8] [--steps 30] [--graph]      (--graph: the pipelines replay a hipGraph per batch, round 6)
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, nprocs, steps, q, barrier, graph=False):
    import torch
    from rangedet_amd import lib as rdlib, synth
    from rangedet_amd.pipeline import InterleavedPipelines
    P = synth.make_weights(seed=18)
    multi = InterleavedPipelines(P, n=2, dtype=rdlib.RD_BF16, wnms_cap=8192, batch=8, graph=graph)
    fr = synth.make_batch(list(range(8)), lib=multi.pipes[0].lib, alloc=multi.pipes[0].alloc)
    for _ in range(4):
        multi.enqueue(fr)
    torch.cuda.synchronize()
    barrier.wait()
    # keep at most two steps queued (like bench.py: a pipeline is read back before it is reused) so that the queue depth, not
    # the GPU, never throttles the enqueue calls: time only the enqueue calls themselves
    spent = 0.0
    t_all = time.perf_counter()
    for i in range(steps):
        t0 = time.perf_counter()
        multi.enqueue(fr)
        spent += time.perf_counter() - t0
        if i % 2 == 1:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_all
    q.put((rank, spent / steps * 1e3, wall / steps * 1e3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--graph", action="store_true")
    a = ap.parse_args()
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    for n in sorted({1, a.procs}):
        q, bar = ctx.Queue(), ctx.Barrier(n)
        ps = [ctx.Process(target=worker, args=(r, n, a.steps, q, bar, a.graph)) for r in range(n)]
        for p in ps:
            p.start()
        res = sorted(q.get(timeout=600) for _ in ps)
        for p in ps:
            p.join()
        enq = [r[1] for r in res]
        print(("hipGraph replay, " if a.graph else "") + "%d process(es): CPU enqueue per 8-frame step min %.2f / mean %.2f / max %.2f ms (%d host threads); wall %.1f ms/step"
              " on the shared GPU" % (n, min(enq), sum(enq) / n, max(enq), os.cpu_count(), sum(r[2] for r in res) / n))


if __name__ == "__main__":
    main()
