"""Frame-parallel multi-GPU execution: one process per GPU, frame f -> rank f % world, weights replicated, NO data-path
collective -- the reference's eval driver does the same with one DetModule per GPU fed from a shared queue
(tools/test.py:117-161).  The only exchange is the gather of the final detections, one `all_gather` of a fixed-size padded
record per step (RCCL over xGMI on the GPUs: latency-only, 77 KB per rank for 8 frames).  The pack (two strided device copies) runs
behind the batch's NMS on its post-processing stream; the COLLECTIVE runs on a communication stream of its own (round 6): no launch
stream ever carries a collective, so a rank that arrives late stalls only the peers' communication streams, never the kernels of the
batches behind it.  The host issues the collective once the pack's event has FIRED (`gather()`), in step order on every rank, instead of
parking a cross-stream wait on that stream when the batch is enqueued: a barrier packet that sits in a hardware queue for the 8 - 24 ms
of a forward costs 7 % of the GPU's throughput by itself (third session of round 6, profiles/r06o_gather_path_cost_bisect.txt).

    shard    = FrameSharding(rank, world)             which frames are mine / where a gathered record belongs
    gatherer = DetectionGather(post, shard, alloc)    pack (rd_copy_rows) + all_gather of a BatchPostProcessor's results
    gatherer.pack(stream) ... (the pack's event has fired) ... gatherer.gather(comm_stream)   |   gatherer.enqueue(stream, comm_stream)
    frames = gatherer.unpack()     # {global frame index: (rows (M,12), M)} on every rank

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
        if backend == "nccl":
            # the collective's kernel runs on the process group's own stream: make that a HIGH-priority queue.  The persistent conv
            # workgroups of the batches in flight fill every CU's LDS, so an RCCL kernel gets on the GPU only when one of them exits --
            # a high-priority queue is served first at that moment instead of competing with three launch streams' next kernels
            try:
                kw["pg_options"] = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            except Exception:      # noqa: BLE001  (an older torch: the default stream priority)
                pass
        dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()


def bind_cpus(local_rank, local_world, max_threads=8):
    """Per-rank CPU affinity, the counterpart of the reference's utils/cpu_affinity.py (bind_cpus_on_ecos :38-47, simple_bind_cpus
    :7-15): the CPUs this process is ALLOWED to run on (cgroup / taskset aware) are cut into local_world contiguous slices and rank r
    keeps slice r -- contiguous CPU numbers share a socket / NUMA node on the usual two-socket GPU hosts, so a rank's host threads
    (enqueue thread, torch's intra-op pool, pinned-buffer copies) stay next to the memory they touch and the 8 ranks of a node do not
    migrate over each other.  RD_NO_AFFINITY=1 leaves the affinity alone.  Returns the CPU list (or None)."""
    if local_world <= 1 or os.environ.get("RD_NO_AFFINITY") or not hasattr(os, "sched_getaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    per = len(allowed) // local_world
    if per < 1:
        return None
    mine = allowed[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    try:
        import torch
        torch.set_num_threads(max(1, min(max_threads, len(mine))))
    except ImportError:
        pass
    return mine


def select_device(local_rank, local_world):
    """The GPU of this rank, safe under HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES: a launcher that gives every process ONE visible
    device (index 0 everywhere) and one that shows every process all of them (index = local rank) both work; anything in between
    is an error, not a silent sharing of GPUs."""
    import torch
    n = torch.cuda.device_count()
    if n >= local_world:
        idx = local_rank
    elif n == 1:
        idx = 0
    else:
        raise RuntimeError("rank %d of %d local ranks sees %d GPUs (HIP_VISIBLE_DEVICES=%r): need one per rank or all of them" %
                           (local_rank, local_world, n, os.environ.get("HIP_VISIBLE_DEVICES")))
    torch.cuda.set_device(idx)
    return torch.device("cuda", idx)


def device_identity(device):
    """16 bytes that identify the physical GPU behind `device` (its UUID; the PCI bus id where the runtime has no UUID)."""
    import hashlib
    import torch
    p = torch.cuda.get_device_properties(device)
    # the PCI address (domain : bus : device) is unique per GPU of a node; the UUID too where its str() is the UUID itself (some torch
    # builds print the wrapper object with its ADDRESS instead, which would make any two processes look distinct: left out then)
    uu = str(getattr(p, "uuid", ""))
    pci = [getattr(p, a, None) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
    ident = "|".join([uu if " object at " not in uu else ""] + ["" if v is None else str(v) for v in pci])
    if not ident.strip("|"):
        ident = "%s/%d" % (p.name, device.index or 0)
    return hashlib.sha256((ident + "@" + os.uname().nodename).encode()).digest()[:16]


def check_ranks(ranks_seen, identities, world):
    """Loud failure when the job is not `world` distinct ranks on `world` distinct GPUs: ranks_seen must be 0 .. world-1 exactly and
    no two ranks may report the same device identity (two processes time-slicing one GPU would still print a throughput)."""
    if list(ranks_seen) != list(range(world)):
        raise RuntimeError("multi-GPU run: the communicator reports ranks %s, expected 0..%d" % (list(ranks_seen), world - 1))
    ids = [bytes(i) for i in identities]
    if len(set(ids)) != len(ids):
        dup = sorted(r for r, i in enumerate(ids) if ids.count(i) > 1)
        raise RuntimeError("multi-GPU run: ranks %s share a GPU (one process per GPU is the contract)" % dup)


class HostCopyLib:
    """rd_copy_rows for host memory: what DetectionGather's pack needs when the process group is gloo and the buffers are numpy
    arrays (CPU test tier, `RD_BENCH_DRYRUN`).  Same arguments as the C ABI entry point (include/rangedet_hip.h)."""

    @staticmethod
    def call(name, src, src_row_bytes, dst, dst_row_bytes, dst_offset_bytes, nbytes, rows, stream=None):
        import ctypes
        assert name == "rd_copy_rows"
        for r in range(rows):
            ctypes.memmove(dst + r * dst_row_bytes + dst_offset_bytes, src + r * src_row_bytes, nbytes)
        return 0


class HostAlloc:
    """numpy-backed stand-in for runtime.TorchAllocator on the gloo path (no GPU)."""
    stream = None

    @staticmethod
    def alloc(nbytes, zero=False):
        return np.zeros(max(int(nbytes), 16) + 64, dtype=np.uint8)

    @staticmethod
    def ptr(buf):
        return buf.ctypes.data

    @staticmethod
    def sync():
        pass


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
        self._work = None

    def _as_torch(self, buf, shape):
        import torch
        if isinstance(buf, np.ndarray):
            return torch.from_numpy(buf[: int(np.prod(shape)) * 4].view(np.float32).reshape(shape))
        return self.A.view_f32(buf, shape)

    def pack(self, stream=None):
        """The two strided copies into the contiguous record buffer, on `stream` (the batch's post-processing stream, behind its NMS)."""
        A, L, p = self.A, self.L, self.post
        st = A.stream_ptr(stream) if hasattr(A, "stream_ptr") else A.stream
        L.call("rd_copy_rows", A.ptr(p.out), p.cap * 48, A.ptr(self.src), self.rec * 4, 0, self.nrow * 48, self.B, st)
        L.call("rd_copy_rows", A.ptr(p.nkeep), 4, A.ptr(self.src), self.rec * 4, self.max_det * 48, 4, self.B, st)

    def gather(self, comm_stream=None):
        """The ONE collective, enqueued on `comm_stream` (None: the current stream) WITHOUT any stream dependency: call it once the pack is
        known to be complete -- the host has waited for (or polled) an event recorded behind `pack()`.  This is the form the timed pipeline
        uses (bench.py): a communication stream that WAITS for the pack's event from the moment the batch is enqueued parks a barrier
        packet in a hardware queue for the whole forward (8 - 24 ms), and that costs 7 % of the GPU's throughput whatever the collective
        is (profiles/r06o_gather_path_cost_bisect.txt); issued after the fact the wait does not exist.  Every rank must issue its gathers in
        the same (step, class) order.  Returns the stream whose completion covers the gathered buffer."""
        import torch.distributed as dist
        A = self.A
        src = self._as_torch(self.src, (self.B * self.rec,))
        dst = self._as_torch(self.dst, (self.shard.world * self.B * self.rec,))
        if not hasattr(A, "torch"):
            # host buffers (gloo): returns at once, unpack() waits for THIS gather only
            self._work = dist.all_gather_into_tensor(dst, src, async_op=True)
            return comm_stream
        if comm_stream is not None:
            with A.torch.cuda.stream(comm_stream):
                dist.all_gather_into_tensor(dst, src)
        else:
            dist.all_gather_into_tensor(dst, src)
        return comm_stream

    def enqueue(self, stream=None, comm_stream=None):
        """pack on `stream` (the batch's post-processing stream; None = current), then the collective on `comm_stream` behind the pack's
        event (None: on `stream` itself -- host buffers / gloo, or a caller with a single stream) -- everything enqueued at once, for a
        caller that cannot come back later.  Returns the stream whose completion covers the gathered buffer (record the batch's done
        event THERE).  On the GPU prefer pack() now and gather() once the pack's event has fired: see gather()."""
        A = self.A
        self.pack(stream)
        if comm_stream is not None and hasattr(A, "torch"):
            A.wait_event(A.record_event(stream), comm_stream)
            return self.gather(comm_stream)
        if stream is not None and hasattr(A, "torch"):
            self.gather(stream)
            return stream
        self.gather(None)
        return stream

    def unpack(self, step=0, sync=True):
        """{global frame index: (rows (M,12) float32, M)} for the B frames of every rank at `step` (host copy)."""
        if self._work is not None:
            self._work.wait()
            self._work = None
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
