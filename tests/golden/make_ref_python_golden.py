#!/usr/bin/env python
"""Golden vectors made by the REFERENCE'S OWN PYTHON, run in the build container only (reads /root/reference; nothing of it
travels: the outputs are data).  Two pins that need neither MXNet nor a GPU ("pins that are available"):

  (a) input_chain_<i>.npz -- the reference's test-time transform chain (config/rangedet/rangedet_veh_wo_aug_4_18e.py:380-399:
      LoadRecord ... TransAndReshape, the classes of rangedet/core/input.py:14-42,89-229,522-624 with the config's own
      parameter classes) applied to rangedet_amd.synth.raw_record(i): every tensor of `data_name` the graph consumes.
      Pins rows a0 / f1 of SURVEY.md section 8 (oracle/input_ref.py and the device kernel rd_input_transform).
  (b) graph_veh_test.json -- the reference's own model code (rangedet/symbol/backbone/dla_backbone.py, meta_kernel.py,
      rangedet/symbol/head/builder.py, mxnext/simple.py, mxnext/complicate.py), executed with rangedet_amd.mx (the recording
      stand-in for the slice of mx.sym the test graph uses) installed as `mxnet`: the recorded test symbol, node for node
      (op, name, attributes, input edges).  Pins the STRUCTURE of rows a2-a6 (layer order, names, kernel / stride / pad /
      no_bias, reshape shapes).  The arithmetic of each MXNet operator remains third-party and unpinned.

Stand-ins installed for imports the reference makes at module scope but never calls on this path: `numba.jit/njit` (identity),
`processing_cxx` (empty module: the test chain does not call it), `mxnet.*` submodules (inert).  No reference source is copied.

  (d) decode_roundtrip.npz -- the reference's own regression-target ENCODER (rangedet/core/input.py:452-507) on chosen boxes / points: the
      inverse of Decode3DBbox, so decode(encode(box)) = box pins the decode's layout and conventions (row a8).

    python tests/golden/make_ref_python_golden.py          # writes tests/golden/input_chain_*.npz, graph_veh_test.json, box_formats.npz,
                                                           #   decode_roundtrip.npz
"""
import importlib
import importlib.abc
import importlib.machinery
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


class _Inert:
    """Placeholder for reference imports that are only touched at import time (base classes, decorators, initialisers)."""
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and isinstance(a[0], type) and not k:
            return a[0]          # used as a class decorator (mx.operator.register("...")(cls), mx.init.register(cls))
        return _Inert()

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Inert()

    def __mro_entries__(self, bases):   # used as a base class
        return (object,)


class _InertModule(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Inert()


class _MxnetSubmodules(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.startswith("mxnet."):
            return importlib.machinery.ModuleSpec(name, self)
        return None

    def create_module(self, spec):
        m = _InertModule(spec.name)
        m.__path__ = []          # (a package, so that mxnet.ndarray.contrib & co. resolve through this finder too)
        return m

    def exec_module(self, module):
        pass


def install_stand_ins():
    from rangedet_amd import mx as rmx

    class RecordingSym:
        """mx.sym / mx.symbol: the recorder's op where it has one; anything else only fails if it is CALLED (mxnext/simple.py
        aliases dozens of operators at import time)."""
        def __getattr__(self, k):
            if not k.startswith("_") and hasattr(rmx, k):
                return getattr(rmx, k)

            def not_on_the_test_path(*a, **kw):
                raise NotImplementedError("mx.sym.%s was called: not recorded by rangedet_amd.mx" % k)
            return not_on_the_test_path

    mx = _InertModule("mxnet")
    mx.__path__ = []
    mx.sym = mx.symbol = RecordingSym()
    class Initializer:           # mx.init.*: the builders only construct them and check isinstance (mxnext/simple.py:140)
        def __init__(self, *a, **k):
            self.args = (a, k)

        def dumps(self):
            return json.dumps([type(self).__name__.lower(), repr(self.args)])
    init = types.SimpleNamespace(Initializer=Initializer, register=lambda c: c)
    for n in ("Constant", "Normal", "One", "Zero", "Xavier", "Uniform", "MSRAPrelu"):
        setattr(init, n, type(n, (Initializer,), {}))
    mx.init = mx.initializer = init

    class EvalMetric:            # base class of rangedet/core/detection_metric.py (training logging, never evaluated here)
        def __init__(self, *a, **k):
            pass
    mx.metric = types.SimpleNamespace(EvalMetric=EvalMetric)
    mx.contrib = types.SimpleNamespace(sym=rmx.contrib, symbol=rmx.contrib)
    mx.sym.__dict__["contrib"] = rmx.contrib
    sys.modules["mxnet"] = mx
    sys.meta_path.insert(0, _MxnetSubmodules())
    nb = types.ModuleType("numba")

    def deco(*a, **k):
        return a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f)
    nb.jit = nb.njit = deco
    sys.modules["numba"] = nb
    sys.modules["processing_cxx"] = types.ModuleType("processing_cxx")
    sys.path.insert(0, REF)


def reference_config():
    cfg = importlib.import_module("config.rangedet.rangedet_veh_wo_aug_4_18e")
    return cfg.get_config(False)


def fixture_record(i, H, W):
    """Synthetic raw record i with, on top of its random runs of missing returns, one solid block of them: its interior pixels
    have no valid return two pixels away in any direction (filled with [80, 0, 0, -1], input.py:124-131), its rim and the
    runs' inner pixels are "car window" pixels (zeroed, input.py:132-135)."""
    from rangedet_amd import synth
    rec = synth.raw_record(i, H=H, W=W)
    h0, w0 = (3 + 2 * i) % (H - 8), (17 + 40 * i) % (W - 12)
    rec["range_image"][h0:h0 + 7, w0:w0 + 11] = -1
    rec["pc_vehicle_frame"][h0:h0 + 7, w0:w0 + 11] = 0
    return rec


def run_reference_chain(transform, rec, tmpdir, pad_hw):
    """The reference's own transform objects, in the config's order, on one record (through its npz / pc_url path)."""
    path = os.path.join(tmpdir, "rec.npz")
    np.savez(path, **rec)
    r = dict(pc_url=path, gt_class=np.array([1.0]), gt_bbox_imu=np.zeros((1, 8, 3)), gt_bbox_csa=np.zeros((1, 7)),
             gt_bbox_yaw=np.zeros(1), points_in_box=np.zeros(1), meta_data=np.zeros((1, 4)))
    for t in transform:
        if type(t).__name__ == "PadData":          # the config's parameter object says (64, 2656): the fixture's own size
            t.pad_short, t.pad_long = pad_hw
        t.apply(r)
    return r


KEYS = ["input_data", "coord_s1"] + ["%s_s%d" % (n, s) for n in ("pc_vehicle_frame", "range_image_mask") for s in (1, 2, 4)]


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs %s (build container only)" % REF)
    import hashlib
    install_stand_ins()
    cfg = reference_config()
    transform, data_name = cfg[9], cfg[10]
    assert [type(t).__module__ for t in transform] == ["rangedet.core.input"] * len(transform)
    assert sys.modules["rangedet.core.input"].__file__.startswith(REF)
    assert set(KEYS) <= set(data_name), data_name
    # (a) small frames as data, one full-size frame as digests
    with tempfile.TemporaryDirectory() as td:
        for i, (H, W, Wp) in enumerate([(16, 250, 256), (16, 250, 256), (12, 126, 128)]):
            rec = fixture_record(i, H, W)
            r = run_reference_chain(transform, rec, td, (H, Wp))
            ri = rec["range_image"][..., 0]
            out = {k: np.asarray(r[k], dtype=np.float32) for k in KEYS}
            np.savez_compressed(os.path.join(HERE, "input_chain_%d.npz" % i), raw_range_image=rec["range_image"],
                                raw_pc_vehicle_frame=rec["pc_vehicle_frame"], raw_inclination=rec["inclination"],
                                raw_azimuth=rec["azimuth"], pad_hw=np.array([H, Wp]), **out)
            print("input_chain_%d: %dx%d (pad %d), %d missing returns" % (i, H, W, Wp, int((ri == -1).sum())))
        from rangedet_amd import synth
        digests = {}
        for i in range(2):
            r = run_reference_chain(transform, synth.raw_record(i), td, (64, 2656))
            digests["raw_record(%d)" % i] = {k: [list(np.asarray(r[k]).shape),
                                                  hashlib.sha256(np.ascontiguousarray(r[k], dtype=np.float32).tobytes()).hexdigest()]
                                             for k in KEYS}
        json.dump(digests, open(os.path.join(HERE, "input_chain_fullsize_sha256.json"), "w"), indent=1)
    # (a') what the reference's Bbox3dAssigner (training chain, rangedet/core/input.py:276-320) hands to assign3D_v2 after
    # LoadRecord + ProcessMissValue: the point array and the mask, captured through a recording stand-in for the compiled call
    RI = sys.modules["rangedet.core.input"]
    captured = {}

    def capture_assign3d(pc, bbox, center, radius, mask, nlz, *lim):
        captured.update(pc=np.array(pc, np.float32), mask=np.array(mask, np.float32), nbox=len(bbox), lim=[float(v) for v in lim])
        return np.full((pc.shape[0],), -1.0, np.float32)
    sys.modules["processing_cxx"].assign3D_v2 = capture_assign3d
    with tempfile.TemporaryDirectory() as td:
        from rangedet_amd import synth
        rec = synth.raw_record(1)
        path = os.path.join(td, "rec.npz")
        np.savez(path, **rec)
        gt = synth.gt_boxes(3, seed=7) if hasattr(synth, "gt_boxes") else None
        if gt is None:
            rng = np.random.default_rng(7)
            c = rng.uniform(-30, 30, (3, 1, 3)).astype(np.float32)
            gt = c + rng.uniform(-2, 2, (3, 8, 3)).astype(np.float32)
        r = dict(pc_url=path, gt_class=np.ones(3), gt_bbox_imu=gt, gt_bbox_csa=np.zeros((3, 7)), gt_bbox_yaw=np.zeros(3),
                 points_in_box=np.zeros(3), meta_data=np.zeros((3, 4)))
        for t in (RI.LoadRecord(), RI.LoadGTInfo(), RI.FilterGTClass([1]), RI.ProcessMissValue(),
                  RI.Bbox3dAssigner(type("P", (), dict(feat_size=(64, 2650))))):
            t.apply(r)
    json.dump({"record": "synth.raw_record(1)", "gt_seed": 7, "nbox": captured["nbox"], "lim": captured["lim"],
               "pc_sha256": hashlib.sha256(captured["pc"].tobytes()).hexdigest(), "pc_shape": list(captured["pc"].shape),
               "mask_sha256": hashlib.sha256(captured["mask"].tobytes()).hexdigest(), "mask_shape": list(captured["mask"].shape),
               "mask_sum": float(captured["mask"].sum())},
              open(os.path.join(HERE, "assigner_state_sha256.json"), "w"), indent=1)
    np.save(os.path.join(HERE, "assigner_gt.npy"), gt)
    print("assigner_state: pc %s mask sum %d" % (captured["pc"].shape, captured["mask"].sum()))
    # (b) the test symbol the reference's builders recorded
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from graph_json import graph_to_json
    sym = cfg[6].test_symbol
    for mod in ("rangedet.symbol.backbone.dla_backbone", "rangedet.symbol.backbone.meta_kernel", "rangedet.symbol.head.builder",
                "mxnext.simple", "mxnext.complicate"):
        assert sys.modules[mod].__file__.startswith(REF), mod
    nodes = graph_to_json(sym)
    json.dump({"made_by": "tests/golden/make_ref_python_golden.py: /root/reference model code recorded by rangedet_amd.mx",
               "config": "config/rangedet/rangedet_veh_wo_aug_4_18e.py get_config(is_train=False)", "nodes": nodes},
              open(os.path.join(HERE, "graph_veh_test.json"), "w"))
    # (c) the eval driver's two box-format helpers (tools/test.py:43-81).  The module loads a compiled MXNet plugin through
    # ctypes at import time; that one call is made inert for the import, nothing else of the module runs (its loop is under
    # `if __name__ == "__main__"`).
    import ctypes
    real_cdll = ctypes.CDLL
    sys.modules["processing_cxx"].wnms_4c = None      # (imported by name at module scope, not called by the helpers)
    ctypes.CDLL = lambda *a, **k: None
    try:
        ref_test = importlib.import_module("tools.test")
    finally:
        ctypes.CDLL = real_cdll
    assert ref_test.__file__.startswith(REF)
    rng = np.random.default_rng(43)
    from rangedet_amd import synth
    d = synth.cluster_dets(12, 9, seed=5)                               # (K,12): 4 corners, yaw, z0, height, score
    b10 = np.concatenate([d[:, :8], d[:, 9:10], d[:, 9:10] + d[:, 10:11]], 1).astype(np.float32)
    b10 = b10[rng.permutation(b10.shape[0])]
    b11 = ref_test.bbox3d_10dim_to_11dim(b10)
    b12 = np.concatenate([b11, rng.uniform(0.5, 1, (b11.shape[0], 1)).astype(np.float32)], 1).astype(np.float32)
    b8 = ref_test.bbox3d_12dim_to_8dim(b12)
    np.savez_compressed(os.path.join(HERE, "box_formats.npz"), b10=b10, b11=b11, b12=b12, b8=b8)
    print("box_formats.npz: %d boxes, 10->11 %s %s, 12->8 %s %s" % (b10.shape[0], b11.shape, b11.dtype, b8.shape, b8.dtype))
    # (d) the reference's regression-TARGET encoder (rangedet/core/input.py:452-507, GenerateTarget.get_rpn_reg_target: per point the 8
    # numbers the box head regresses -- sqrt-compressed offsets in the point's azimuth frame, log w, log l, cos / sin of the relative yaw,
    # bottom height, log h) on boxes and points of our choice.  Decode3DBbox (operator_cxx/contrib/decode_3d_bbox-inl.h:170-262, needs
    # MXNet to build) is its inverse: decode(encode(box, point), point) must give the box back -- the pin for the decode's delta layout,
    # frame conventions and corner order that needs neither MXNet nor a GPU.
    rng = np.random.default_rng(91)
    M, PP = 40, 12
    gt = np.stack([rng.uniform(-70, 70, M), rng.uniform(-70, 70, M), rng.uniform(-1.0, 2.5, M), rng.uniform(3.2, 12.0, M), rng.uniform(1.5, 3.0, M),
                   rng.uniform(1.3, 3.5, M), rng.uniform(-np.pi, np.pi, M)], 1).astype(np.float32)            # x, y, z, l, w, h, yaw
    off = rng.uniform(-0.5, 0.5, (M, PP, 3)) * gt[:, None, 3:6]                                                # inside the box, box frame
    c, s_ = np.cos(gt[:, 6])[:, None], np.sin(gt[:, 6])[:, None]
    pts = np.stack([gt[:, None, 0] + off[..., 0] * c - off[..., 1] * s_, gt[:, None, 1] + off[..., 0] * s_ + off[..., 1] * c,
                    gt[:, None, 2] + off[..., 2]], 2).astype(np.float32)                                        # (M, PP, 3)
    ind = np.repeat(np.arange(M), PP).astype(np.int64)
    pts = pts.reshape(-1, 3)
    pts[::37] = rng.uniform(-70, 70, (len(pts[::37]), 3)).astype(np.float32)                                    # a few background points
    ind[::37] = -1
    Hh, Ww = 16, (M * PP) // 16
    enc = object.__new__(RI.GenerateTarget)
    tgt = RI.GenerateTarget.get_rpn_reg_target(enc, pts.reshape(Hh, Ww, 3), gt, ind.reshape(Hh, Ww, 1))
    assert tgt.shape == (M * PP, 8) and not np.any(tgt[ind == -1])
    np.savez_compressed(os.path.join(HERE, "decode_roundtrip.npz"), pc=pts, gt=gt, ind=ind, deltas=np.asarray(tgt, np.float32),
                        deltas_dtype=str(tgt.dtype))
    print("decode_roundtrip.npz: %d points of %d boxes, targets %s %s" % (len(pts), M, tgt.shape, tgt.dtype))
    ops = {}
    for n in nodes:
        ops[n["op"]] = ops.get(n["op"], 0) + 1
    print("graph_veh_test.json: %d nodes" % len(nodes), ops)


if __name__ == "__main__":
    main()
