"""Frame-parallel multi-GPU execution: one process per GPU, frame f -> rank f % world, weights replicated, NO data-path
collective -- the reference's eval driver does the same with one DetModule per GPU fed from a shared queue
(tools/test.py:117-161).  The only exchange is the gather of the final detections, one `all_gather` of a fixed-size padded
record per step (RCCL over xGMI on the GPUs: latency-only, 77 KB per rank for 8 frames), enqueued on the batch's
post-processing stream behind its NMS so that no launch stream ever waits for it.

    shard    = FrameSharding(rank, world)             which frames are mine / where a gathered record belongs
    gatherer = DetectionGather(post, shard, alloc)    pack (rd_copy_rows) + all_gather of a BatchPostProcessor's results
    gatherer.enqueue(stream)   ...   frames = gatherer.unpack()     # {global frame index: (rows (M,12), M)} on every rank

The backend is whatever process group is initialised ("nccl" == RCCL on the GPUs; "gloo" in the CPU test tier, where the
buffers are host memory and the same code runs).
"""
import os

import numpy as np

MAX_DET = 200     # rpn_post_nms_top_n (config:139): rows of a frame's padded record


class FrameSharding:
    """frame f -> rank f % world; step s of rank r with B frames per step handles frames r + world * (s*B + j), j < B."""

    def __init__(self, rank=None, world=None):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        assert 0 <= self.rank < self.world

    def owner(self, frame):
        return frame % self.world

    def frames_of_step(self, step, batch, rank=None):
        r = self.rank if rank is None else rank
        return [r + self.world * (step * batch + j) for j in range(batch)]

    def mine(self, nframes):
        """The frames of range(nframes) this rank owns, in processing order."""
        return list(range(self.rank, nframes, self.world))

    def steps(self, nframes, batch):
        """Number of steps every rank runs so that all of range(nframes) is covered (the last ones may be padded)."""
        per_rank = -(-nframes // self.world)
        return -(-per_rank // batch)


def record_floats(max_det=MAX_DET):
    """One frame's record: max_det (M,12) rows + one word holding the int32 keep count (bit pattern)."""
    return max_det * 12 + 1


def init_process_group(backend=None, device=None):
    """torch.distributed from the launcher's environment (RANK / WORLD_SIZE / MASTER_*: torch.distributed.run sets them).
    nccl (== RCCL) when a GPU device is given, else gloo.  Returns (rank, world)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC (the host driver supports nothing else)
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if device is not None else "gloo"
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()


class DetectionGather:
    """Padded per-frame records of one BatchPostProcessor (B frames) gathered from every rank.

    pack: two strided device copies (rd_copy_rows) -- the first max_det rows of each frame's (cap,12) result and its keep
    count -- into one contiguous (B, record_floats) buffer; gather: ONE all_gather into (world, B, record_floats)."""

    def __init__(self, post, shard, alloc, lib, max_det=MAX_DET):
        self.post, self.shard, self.A, self.L, self.max_det = post, shard, alloc, lib, max_det
        self.B, self.rec = post.B, record_floats(max_det)
        self.nrow = min(max_det, post.cap)
        self.src = alloc.alloc(self.B * self.rec * 4, zero=True)
        self.dst = alloc.alloc(shard.world * self.B * self.rec * 4, zero=True)

    def _as_torch(self, buf, shape):
        import torch
        if isinstance(buf, np.ndarray):
            return torch.from_numpy(buf[: int(np.prod(shape)) * 4].view(np.float32).reshape(shape))
        return self.A.view_f32(buf, shape)

    def enqueue(self, stream=None):
        """On `stream` (the batch's post-processing stream; None = current): pack, then the collective."""
        import torch.distributed as dist
        A, L, p = self.A, self.L, self.post
        st = A.stream_ptr(stream) if hasattr(A, "stream_ptr") else A.stream
        L.call("rd_copy_rows", A.ptr(p.out), p.cap * 48, A.ptr(self.src), self.rec * 4, 0, self.nrow * 48, self.B, st)
        L.call("rd_copy_rows", A.ptr(p.nkeep), 4, A.ptr(self.src), self.rec * 4, self.max_det * 48, 4, self.B, st)
        src = self._as_torch(self.src, (self.B * self.rec,))
        dst = self._as_torch(self.dst, (self.shard.world * self.B * self.rec,))
        if stream is not None and hasattr(A, "torch"):
            with A.torch.cuda.stream(stream):
                dist.all_gather_into_tensor(dst, src)
        else:
            dist.all_gather_into_tensor(dst, src)

    def unpack(self, step=0, sync=True):
        """{global frame index: (rows (M,12) float32, M)} for the B frames of every rank at `step` (host copy)."""
        if sync:
            self.A.sync()
        buf = self._as_torch(self.dst, (self.shard.world, self.B, self.rec))
        a = np.array(buf.cpu().numpy() if hasattr(buf, "cpu") else buf)
        out = {}
        for r in range(self.shard.world):
            for j, f in enumerate(self.shard.frames_of_step(step, self.B, rank=r)):
                M = int(a[r, j, -1:].view(np.int32)[0])
                out[f] = (a[r, j, : min(M, self.nrow) * 12].reshape(-1, 12).copy(), M)
        return out
