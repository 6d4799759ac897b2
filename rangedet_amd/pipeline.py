"""End-to-end frame runner: the build's counterpart of the per-frame body of the reference's eval driver
(tools/test.py:143-161 forward + :184-224 post-process): config -> test symbol -> lowered plan -> Executor, then
score filter, 10->11 dim, weighted NMS and 12->8 dim on the device.
"""
import os
import numpy as np

from . import devswitch, lib as rdlib
from .config import rangedet_veh_wo_aug_4_18e as cfgmod
from .lower import lower
from .runtime import Executor, TorchAllocator


def input_shapes(H, W, strides=(1, 2, 4), channels=8):
    shapes = {'input_data': (channels, H, W), 'coord_s1': (3, H, W)}
    for s in strides:
        shapes['pc_vehicle_frame_s%d' % s] = (H * W // s, 3)
        shapes['range_image_mask_s%d' % s] = (H * W // s,)
    return shapes



def _stream_prio(var):
    """Development switch: queue priority of the launch / post-processing streams, 0 = normal (default), -1 = high."""
    return int(devswitch.get(var, "0"))


class BatchPostProcessor:
    """The same post-processing for all frames of a batch in ONE set of launches (rd_*_batched): the greedy NMS scan is a
    single latency-bound wavefront per frame, so B of them run side by side instead of back to back.  Per-frame buffers
    are slices of contiguous allocations; frame b's results are read back with collect(b)."""

    def __init__(self, B, k, min_score, thr_lo, thr_hi, is_3d_iou, lib, alloc, cap=4096, hash_scale=100, tie_order="reference", wnms_diag=0):
        self.B, self.k, self.cap = B, k, min(cap, k, rdlib.RD_WNMS_MAX_K)
        self.hash_scale, self.tie_order = int(hash_scale), tie_order
        self.wnms_diag = int(wnms_diag)     # rdlib.RD_WNMS_DIAG_* bits (test aids: same results through other code paths)
        self.min_score, self.thr_lo, self.thr_hi, self.is3d = min_score, thr_lo, thr_hi, int(is_3d_iou)
        self.L, self.A = lib, alloc
        A, L = alloc, lib
        self.dets = A.alloc(B * k * 12 * 4)
        self.count = A.alloc(max(16, 4 * B), zero=True)
        self.ws_f_bytes = L.raw("rd_score_filter_workspace_bytes")(k) * B
        self.ws_f = A.alloc(self.ws_f_bytes)
        self.ws_w_bytes = L.raw("rd_wnms_workspace_bytes")(self.cap) * B
        self.ws_w = A.alloc(self.ws_w_bytes)
        self.out = A.alloc(B * self.cap * 12 * 4)
        self.keep = A.alloc(B * self.cap * 4)
        self.nkeep = A.alloc(max(16, 4 * B), zero=True)
        self.out8 = A.alloc(B * self.cap * 8 * 4)
        # tie_order "stable": the rows reaching WNMS are already in (score desc, index asc) order -- get_sorted_foreground
        # sorts and the score filter is a stable compaction -- so the processing order is the identity.  "reference"
        # (default): the library replays the reference's std::sort (nms.h:786-792) so tied scores are processed in its order
        self.identity = A.upload(np.arange(self.cap, dtype=np.int32))
        self._host = None          # one host copy of the batch's results per sync (collect_all)

    def enqueue_filter(self, score_ptr, score_bs, box_ptr, box_bs, stream=None):
        L, A = self.L, self.A
        st = A.stream_ptr(stream) if hasattr(A, "stream_ptr") else A.stream
        L.call("rd_score_filter_dets_batched", score_ptr, score_bs, box_ptr, box_bs, self.k, self.min_score, A.ptr(self.dets),
               self.k * 12, A.ptr(self.count), A.ptr(self.ws_f), self.ws_f_bytes, self.B, st)

    def enqueue_nms(self, stream=None):
        L, A = self.L, self.A
        st = A.stream_ptr(stream) if hasattr(A, "stream_ptr") else A.stream
        order = A.ptr(self.identity) if self.tie_order == "stable" else None
        L.call("rd_wnms_4c_batched", A.ptr(self.dets), self.k * 12, self.cap, A.ptr(self.count), order, 0,
               rdlib.RD_TIE_REFERENCE | self.wnms_diag, self.thr_lo, self.thr_hi, self.is3d, self.hash_scale, A.ptr(self.out), self.cap * 12,
               A.ptr(self.keep), self.cap, A.ptr(self.nkeep), A.ptr(self.ws_w), self.ws_w_bytes, self.B, st)
        L.call("rd_dets12_to_8_batched", A.ptr(self.out), self.cap * 12, self.cap, A.ptr(self.nkeep), A.ptr(self.out8),
               self.cap * 8, self.B, st)

    def collect_all(self, done=None):
        """Results of all B frames with ONE wait and ONE host copy of each array: `done` = an event recorded behind this
        batch's post-processing on its stream (only that event is waited for, so other batches in flight keep running);
        None = device-wide sync."""
        A = self.A
        if done is not None and hasattr(done, "synchronize"):
            done.synchronize()
        else:
            A.sync()
        count = np.array(A.to_numpy(A.view_i32(self.count, (self.B,))))
        nkeep = np.array(A.to_numpy(A.view_i32(self.nkeep, (self.B,))))
        if int(count.max()) > self.cap:
            raise rdlib.RangeDetError(rdlib.RD_EWORKSPACE, "%d detections above min_score exceed the WNMS capacity %d" % (int(count.max()), self.cap))
        m = int(nkeep.max()) if self.B else 0
        rows = np.array(A.to_numpy(A.view_f32(self.out, (self.B, self.cap, 12))[:, :m]))
        keep = np.array(A.to_numpy(A.view_i32(self.keep, (self.B, self.cap))[:, :m]))
        d8 = np.array(A.to_numpy(A.view_f32(self.out8, (self.B, self.cap, 8))[:, :m]))
        return [dict(num_candidates=int(count[b]), wnms_rows=rows[b, :nkeep[b]].copy(), keep_inds=keep[b, :nkeep[b]].copy(),
                     det_xyzlwhyaws=d8[b, :nkeep[b]].copy()) for b in range(self.B)]

    def collect(self, b=0):
        return self.collect_all()[b]


class Nms3dPostProcessor:
    """The harness steps after the graph when RpnParam.wnms is False (tools/test.py:193-224, `not pTest.nms.wnms`): the graph
    already ran contrib.NMS3D, so what is left is: scores of the kept boxes, score filter + 10->11 dim, 12->8 dim.  Same
    collect() result as BatchPostProcessor (wnms_rows = the filtered (K,12) rows; keep_inds = their NMS3D indices)."""

    def __init__(self, B, k, max_keep, min_score, lib, alloc):
        self.B, self.k, self.mk, self.min_score, self.L, self.A = B, k, max_keep, min_score, lib, alloc
        A, L = alloc, lib
        self.cap = max_keep
        self.kscore = A.alloc(B * max_keep * 4)
        self.dets = A.alloc(B * max_keep * 12 * 4)
        self.count = A.alloc(max(16, 4 * B), zero=True)
        self.ws_bytes = L.raw("rd_score_filter_workspace_bytes")(max_keep) * B
        self.ws = A.alloc(self.ws_bytes)
        self.out8 = A.alloc(B * max_keep * 8 * 4)
        self._keep = None

    def enqueue(self, score_ptr, score_bs, final_ptr, keep_ptr, keep_view, stream=None):
        L, A, mk = self.L, self.A, self.mk
        st = A.stream_ptr(stream) if hasattr(A, "stream_ptr") else A.stream
        self._keep = keep_view
        L.call("rd_gather_keep_scores", score_ptr, score_bs, self.k, keep_ptr, mk, A.ptr(self.kscore), self.B, st)
        L.call("rd_score_filter_dets_batched", A.ptr(self.kscore), mk, final_ptr, mk * 10, mk, self.min_score, A.ptr(self.dets),
               mk * 12, A.ptr(self.count), A.ptr(self.ws), self.ws_bytes, self.B, st)
        L.call("rd_dets12_to_8_batched", A.ptr(self.dets), mk * 12, mk, A.ptr(self.count), A.ptr(self.out8), mk * 8, self.B, st)

    def collect_all(self, done=None):
        if done is not None and hasattr(done, "synchronize"):
            done.synchronize()
        else:
            self.A.sync()
        return [self._collect(b) for b in range(self.B)]

    def collect(self, b=0):
        self.A.sync()
        return self._collect(b)

    def _collect(self, b):
        A = self.A
        K = int(A.to_numpy(A.view_i32(self.count, (self.B,)))[b])
        rows = A.to_numpy(A.view_f32(self.dets, (self.B, self.mk, 12)))[b, :K].copy()
        d8 = A.to_numpy(A.view_f32(self.out8, (self.B, self.mk, 8)))[b, :K].copy()
        keep = np.array(A.to_numpy(self._keep))[b]
        sc = A.to_numpy(A.view_f32(self.kscore, (self.B, self.mk)))[b]
        keep = keep[(keep >= 0) & (sc > self.min_score)]
        return dict(num_candidates=K, wnms_rows=rows, keep_inds=keep, det_xyzlwhyaws=d8)


class _FrameView:
    """pipe.post[b]: frame b of the batched post-processor behind the single-frame PostProcessor interface."""

    def __init__(self, bp, b):
        self.bp, self.b = bp, b
        self.cap = bp.cap

    def collect(self):
        return self.bp.collect(self.b)

    def out_rows(self):
        A = self.bp.A
        return A.view_f32(self.bp.out, (self.bp.B, self.bp.cap, 12))[self.b]

    def nkeep_view(self):
        A = self.bp.A
        return A.view_i32(self.bp.nkeep, (self.bp.B,))[self.b:self.b + 1]


class RangeDetPipeline:
    def __init__(self, params, dtype=rdlib.RD_BF16, feat_size=(64, 2650), pad_field=(64, 2656), batch=1,
                 pre_nms_top_n=50000, wnms_cap=4096, variant="veh", lib=None, alloc=None, wnms=True, tie_order="reference",
                 hash_scale=100, wnms_diag=0, graph=False):
        """wnms=False builds the graph with contrib.NMS3D inside (RpnParam.wnms = False, builder.py:530-534) and runs the
        matching harness branch (tools/test.py:193-196) instead of the weighted NMS.
        tie_order: "reference" = rows with equal scores are processed in the order the reference's std::sort leaves them
        (nms.h:786-792, replayed on the device); "stable" = in index order (no extra kernel).  hash_scale: the BBoxHash cell
        size tools/test.py:216 passes (100).  wnms_cap: rows per frame the weighted NMS is sized for; collect() raises when a
        frame had more candidates above min_score (the device never truncates silently: the true count comes back).
        variant "kitti" (BASELINE config 5): 5 input channels, vehicle + pedestrian heads -- one post-processor per class, each
        with its class's top-k and min_score (builder.py:467-478 slices the classes; the reference's driver itself asserts one
        class, tools/test.py:182, so the per-class loop is this harness's: det_xyzlwhyaws keyed by class as tools/test.py:224).
        pre_nms_top_n: an int (first class) or {class: k}."""
        topn = pre_nms_top_n if isinstance(pre_nms_top_n, dict) else {cfgmod.variant_classes(variant)[0]: pre_nms_top_n}
        self.cfg = cfgmod.get_config(False, variant=variant, feat_size=feat_size, pad_field=pad_field,
                                     batch_image=batch, pre_nms_top_n=topn, wnms=wnms)
        self.wnms = bool(wnms)
        General, RpnParam, ModelParam, TestParam = self.cfg[0], self.cfg[2], self.cfg[6], self.cfg[8]
        self.lib = lib or rdlib.get_lib()
        self.alloc = alloc or TorchAllocator()
        self.class_names = tuple(General.class_names)
        nch = cfgmod.KITTI_INPUT_CHANNELS if variant == "kitti" else 8
        self.plan = lower(ModelParam.test_symbol, input_shapes(*pad_field, channels=nch), dtype, batch)
        self.exe = Executor(self.plan, params, self.lib, self.alloc)
        self.ks = {c: RpnParam.all_proposal.rpn_pre_nms_top_n[c] for c in self.class_names}
        self.batch = batch
        # graph=True: the forward + batched post-processing of a batch (~80 launches enqueued one ctypes call at a time) are captured
        # ONCE per set of input buffers into a hipGraph (stream capture through torch.cuda.CUDAGraph) and replayed with one call:
        # the host's enqueue cost per batch drops from ~1 ms to the replay call.  Inputs must be resident float32 device tensors (their
        # addresses are part of the graph); anything else, and input sets beyond `max_graphs`, run through the eager path.
        self.use_graph = bool(graph) and hasattr(self.alloc, "torch")
        self._graphs = {}
        self._capture_stream = None
        self.max_graphs = 8
        self.graph_replays = 0
        self._post_stream = None
        # score filter + NMS on the stream the forward was launched on.  Rounds 1 - 4 used a side stream per pipeline (the NMS of batch i
        # beside the forward of batch i + 1); with two or more batches in flight the other pipelines give that overlap anyway (round 5),
        # and for ONE pipeline alone the side stream loses too (round 6: 846.5 vs 838.6 frames/s, profiles/r06o_single_pipeline_post_stream_ab.txt):
        # its wait for the forward's event sits in a hardware queue for the whole forward, and a parked barrier packet slows every
        # dispatch on the other queues.  RD_POST_SIDE_STREAM=1 (development switch) brings the side stream back.
        self.post_on_launch_stream = not devswitch.get("RD_POST_SIDE_STREAM")
        self._filter_done = None
        self._post_done = None
        assert self.wnms or len(self.class_names) == 1, "the NMS3D branch is single-class (tools/test.py:182)"
        self.bposts = {}
        for c in self.class_names:
            if self.wnms:
                self.bposts[c] = BatchPostProcessor(batch, self.ks[c], TestParam.min_score[c], TestParam.nms.thr_lo,
                                                    TestParam.nms.thr_hi, TestParam.nms.is_3d_iou, self.lib, self.alloc, wnms_cap,
                                                    hash_scale=hash_scale, tie_order=tie_order, wnms_diag=wnms_diag)
            else:
                self.bposts[c] = Nms3dPostProcessor(batch, self.ks[c], RpnParam.all_proposal.rpn_post_nms_top_n[c],
                                                    TestParam.min_score[c], self.lib, self.alloc)
        # (first class: what single-class callers -- bench.py, evaluate, the gather -- use)
        self.k = self.ks[self.class_names[0]]
        self.bpost = self.bposts[self.class_names[0]]
        self.post = [_FrameView(self.bpost, b) for b in range(batch)]

    def forward(self, inputs):
        """Graph outputs only: [rec_id, then per class (fg_cls_score (B,k), decoded_bbox (B,k,10), zeros), gt_bbox_imu, gt_class]."""
        return self.exe.forward(inputs)

    def enqueue(self, inputs):
        """forward on the current stream; score filter + WNMS + 12->8 on a side stream so that the (latency-bound,
        one-CU) greedy scan of frame i overlaps the convolutions of frame i+1.  (graph=True: one hipGraph replay, see _enqueue_graph.)"""
        if self.use_graph:
            outs = self._enqueue_graph(inputs)
            if outs is not None:
                return outs
        return self._enqueue_eager(inputs)

    def _record(self, inputs):
        """The batch's whole launch sequence on the CURRENT stream, no events: forward, then per class score filter + NMS + 12 -> 8."""
        outs = self.exe.forward(inputs)
        ptr = lambda t: self.alloc.ptr(t) if hasattr(t, "data_ptr") else t.ctypes.data
        if not self.wnms:
            self.bpost.enqueue(ptr(outs[1]), self.k, ptr(outs[2]), ptr(outs[3]), outs[3], stream=None)
            return outs
        for ci, c in enumerate(self.class_names):
            sc, bx = outs[1 + 3 * ci], outs[2 + 3 * ci]
            sc_bs = (ptr(sc[1]) - ptr(sc[0])) // 4 if self.batch > 1 else 0
            bx_bs = (ptr(bx[1]) - ptr(bx[0])) // 4 if self.batch > 1 else 0
            self.bposts[c].enqueue_filter(ptr(sc[0]), sc_bs, ptr(bx[0]), bx_bs, stream=None)
        for c in self.class_names:
            self.bposts[c].enqueue_nms(stream=None)
        return outs

    def _enqueue_graph(self, inputs):
        """One hipGraph replay on the current stream (the pipeline's launch stream).  The first batch that arrives with a new set of input
        buffers runs eagerly (which also sets every kernel's per-device attributes and checks the shapes) and is then CAPTURED -- the
        capture records launches, it does not execute them -- for every later batch in the same buffers.  Returns None when the inputs
        cannot be graphed (host arrays, non-contiguous / non-float32 tensors, too many distinct input sets): the caller runs eagerly."""
        A = self.alloc
        t = A.torch
        key = []
        for name in sorted(inputs):
            v = inputs[name]
            if v is None:
                continue
            if not (hasattr(v, "data_ptr") and v.is_cuda and v.is_contiguous() and v.dtype == t.float32):
                return None
            key.append((name, v.data_ptr(), tuple(v.shape)))
        key = tuple(key)
        cur = t.cuda.current_stream(A.device)
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= self.max_graphs:
                return None
            outs = self._enqueue_eager(inputs, side_ok=False)      # this batch: eagerly, everything on the current stream
            g = t.cuda.CUDAGraph()
            if self._capture_stream is None:
                self._capture_stream = t.cuda.Stream(device=A.device)
            cap = self._capture_stream
            cap.wait_stream(cur)
            with t.cuda.graph(g, stream=cap):
                gouts = self._record(inputs)
            cur.wait_stream(cap)
            self._graphs[key] = (g, gouts, dict(inputs))             # (the input tensors stay alive as long as the graph that reads them)
            return outs
        g, outs, _ = ent
        g.replay()
        self.graph_replays += 1
        self._post_stream = cur
        self._filter_done = None
        self._post_done = A.record_event(cur)
        return outs

    def _enqueue_eager(self, inputs, side_ok=True):
        A = self.alloc
        side = hasattr(A, "new_stream")
        if side and (self.post_on_launch_stream or not side_ok):
            # (set by InterleavedPipelines with two or more batches in flight) post-processing on the batch's own launch stream: the
            # next batch on this stream starts behind this batch's NMS while the OTHER pipeline's forward has the GPU -- the overlap the
            # side stream exists for is already there, and two streams fewer compete for workgroup slots: 942.3 vs 939.1 frames/s over
            # nine same-box alternations (profiles/EXPERIMENTS.md, round 5)
            self._post_stream = A.torch.cuda.current_stream(A.device)
        elif side and self._post_stream is None:
            self._post_stream = A.new_stream(priority=_stream_prio("RD_POST_STREAM_PRIO"))
        if side and self._filter_done is not None:
            A.wait_event(self._filter_done)          # previous frame's filter has consumed the score / box buffers
        outs = self.exe.forward(inputs)
        if side:
            A.wait_event(A.record_event(), self._post_stream)
        # one batched score filter per class (it is what reads the graph's score / box buffers: the next batch's forward only
        # has to wait for these short kernels), then one batched weighted NMS per class for all frames
        ptr = lambda t: self.alloc.ptr(t) if hasattr(t, "data_ptr") else t.ctypes.data
        if not self.wnms:   # outs = [rec_id, score (B,k), bbox_after_nms (B,mk,10), keep_inds (B,mk) int32, ...]
            sc, bx = outs[1], outs[2]
            self.bpost.enqueue(ptr(sc), self.k, ptr(bx), ptr(outs[3]), outs[3], stream=self._post_stream)
            self._filter_done = A.record_event(self._post_stream) if side else None
            self._post_done = self._filter_done
            return outs
        for ci, c in enumerate(self.class_names):
            sc, bx = outs[1 + 3 * ci], outs[2 + 3 * ci]
            sc_bs = (ptr(sc[1]) - ptr(sc[0])) // 4 if self.batch > 1 else 0
            bx_bs = (ptr(bx[1]) - ptr(bx[0])) // 4 if self.batch > 1 else 0
            self.bposts[c].enqueue_filter(ptr(sc[0]), sc_bs, ptr(bx[0]), bx_bs, stream=self._post_stream)
        self._filter_done = A.record_event(self._post_stream) if side else None
        for c in self.class_names:
            self.bposts[c].enqueue_nms(stream=self._post_stream)
        self._post_done = A.record_event(self._post_stream) if side else None
        return outs

    def collect(self):
        """Per frame: the first class's result dict (num_candidates, wnms_rows, keep_inds, det_xyzlwhyaws); with several classes
        the same per class under `per_class` -- waits only for this pipeline's own post-processing event."""
        per = {c: self.bposts[c].collect_all(self._post_done) for c in self.class_names}
        res = []
        for b in range(self.batch):
            r = dict(per[self.class_names[0]][b])
            if len(self.class_names) > 1:
                r["per_class"] = {c: per[c][b] for c in self.class_names}
            res.append(r)
        return res

    def run(self, inputs):
        outs = self.enqueue(inputs)
        res = self.collect()
        r0 = dict(res[0]) if self.batch == 1 else dict(frames=res)
        r0["fg_cls_score"], r0["decoded_bbox"] = outs[1], outs[2]
        return r0


class InterleavedPipelines:
    """`n` pipelines, each with its own launch stream (forward AND post-processing of a batch on it): successive batches
    alternate between them, so n batches are in flight.  The persistent conv kernels give every CU a fixed list of
    tiles; when the tile count is not a multiple of the CU count (W = 1328 and W <= 332 levels) the tail of a launch leaves
    CUs idle -- the other batches' launches fill those tails and the ~80 launch gaps per batch (n = 2: +8 % in round 2; n = 3: another
    +2 % since round 5; n = 4 gives nothing more, and needs GPU_MAX_HW_QUEUES > 4 not to lose: profiles/EXPERIMENTS.md, round 6).
    The caller keeps a batch's input tensors alive until its results are collected."""

    def __init__(self, params, n=2, **kw):
        self.pipes = [RangeDetPipeline(params, **kw) for _ in range(n)]
        for p in self.pipes:      # (RD_POST_SIDE_STREAM=1: a side stream per pipeline as in rounds 1 - 4, for A/B runs)
            p.post_on_launch_stream = not devswitch.get("RD_POST_SIDE_STREAM")
        A = self.pipes[0].alloc
        self.streams = [A.new_stream(priority=_stream_prio("RD_LAUNCH_STREAM_PRIO")) for _ in range(n)] if hasattr(A, "new_stream") else [None] * n
        self._i = 0

    def stream_context(self, j):
        import contextlib
        st = self.streams[j]
        return self.pipes[j].alloc.torch.cuda.stream(st) if st is not None else contextlib.nullcontext()

    def enqueue(self, inputs):
        """Enqueue one batch on the next pipeline; returns (pipeline index, graph outputs)."""
        j = self._i % len(self.pipes)
        self._i += 1
        with self.stream_context(j):
            return j, self.pipes[j].enqueue(inputs)

    def collect(self, j):
        """Wait for pipeline j's last batch (its own post-processing event only) and read back its detections: one dict per frame."""
        return self.pipes[j].collect()
