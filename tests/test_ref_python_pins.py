"""Pins made by the REFERENCE'S OWN PYTHON (tests/golden/make_ref_python_golden.py, run in the build container against
/root/reference; only data is committed):

  * input_chain_*.npz, input_chain_fullsize_sha256.json -- outputs of the reference's test-time transform objects
    (rangedet/core/input.py, parameters of config/rangedet/rangedet_veh_wo_aug_4_18e.py:245-399): pins oracle/input_ref.py
    bit for bit (rows a0 / f1) and, through it and directly, the device kernel rd_input_transform;
  * graph_veh_test.json -- the test symbol recorded while the reference's dla_backbone.py / meta_kernel.py / head/builder.py /
    mxnext ran on rangedet_amd.mx: the graph this package's mirror of that code records must be the same node for node
    (structure of rows a2-a6: layer order, names, kernel / stride / pad / no_bias, reshape shapes).
"""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import BOTH
from graph_json import first_difference, graph_to_json
from oracle import input_ref as IR

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ["input_data", "coord_s1"] + ["%s_s%d" % (n, s) for n in ("pc_vehicle_frame", "range_image_mask") for s in (1, 2, 4)]
CHAINS = sorted(glob.glob(os.path.join(GOLD, "input_chain_[0-9].npz")))


def _raw(g):
    return dict(range_image=g["raw_range_image"], pc_vehicle_frame=g["raw_pc_vehicle_frame"], inclination=g["raw_inclination"],
                azimuth=g["raw_azimuth"])


@pytest.mark.parametrize("f", CHAINS, ids=os.path.basename)
def test_oracle_input_chain_equals_reference_python(f):
    g = np.load(f)
    ri = g["raw_range_image"][..., 0]
    assert (ri == -1).sum() > 100                       # runs of missing returns + one solid block (far fill AND car window)
    out = IR.transform(_raw(g), tuple(int(v) for v in g["pad_hw"]))
    for k in KEYS:
        want = g[k]
        got = out[k][0]
        assert got.shape == want.shape, (k, got.shape, want.shape)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (k, np.abs(got - want).max())
    # both kinds of filled pixels are present in the fixture: far fill (range 80 -> clipped / normalised) and zeroed car-window
    H, W = ri.shape
    unnorm_far = (80.0 - 20.0) / 1500.0 ** 0.5
    assert np.isclose(g["input_data"][0][:H, :W], unnorm_far, atol=1e-6).sum() > 0


def test_oracle_input_chain_full_size_digests():
    from rangedet_amd import synth
    d = json.load(open(os.path.join(GOLD, "input_chain_fullsize_sha256.json")))
    for i in range(len(d)):
        out = IR.transform(synth.raw_record(i), (64, 2656))
        for k in KEYS:
            shape, digest = d["raw_record(%d)" % i][k]
            a = np.ascontiguousarray(out[k][0], dtype=np.float32)
            assert list(a.shape) == shape and hashlib.sha256(a.tobytes()).hexdigest() == digest, (i, k)


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_device_input_transform_equals_reference_python(be):
    """rd_input_transform (one launch) against the reference chain's outputs directly: bit-equal; the azimuth channel to 1e-6
    (device atan2f vs numpy's)."""
    from rangedet_amd.input_transform import DeviceInputTransform
    for f in CHAINS:
        g = np.load(f)
        H, Wp = (int(v) for v in g["pad_hw"])
        out = DeviceInputTransform(pad_hw=(H, Wp), lib=be.lib, alloc=be.alloc)([_raw(g)])
        be.alloc.sync()
        for k in KEYS:
            got = np.array(be.alloc.to_numpy(out[k]))[0]
            want = g[k]
            assert got.shape == want.shape, k
            if k == "input_data":
                assert np.array_equal(got[:7], want[:7]) and np.abs(got[7] - want[7]).max() < 1e-6
            else:
                assert np.array_equal(got, want), k


def test_mirror_graph_equals_reference_recorded_graph():
    from rangedet_amd.config import rangedet_veh_wo_aug_4_18e as cfgmod
    ref = json.load(open(os.path.join(GOLD, "graph_veh_test.json")))["nodes"]
    mine = graph_to_json(cfgmod.get_config(False)[6].test_symbol)
    assert first_difference(mine, ref) is None, first_difference(mine, ref)
    ops = [n["op"] for n in ref]
    assert ops.count("Convolution") == 91 and ops.count("Deconvolution") == 4 and ops.count("BatchNorm") == 88


def test_box_format_helpers_equal_reference_python():
    """oracle/cpu_ops.py bbox3d_10dim_to_11dim / bbox3d_12dim_to_8dim vs the reference's own functions (tools/test.py:43-81)."""
    from oracle import cpu_ops
    g = np.load(os.path.join(GOLD, "box_formats.npz"))
    b11 = cpu_ops.bbox3d_10dim_to_11dim(g["b10"])
    assert b11.dtype == g["b11"].dtype and np.array_equal(b11.view(np.uint32), g["b11"].view(np.uint32))
    b8 = cpu_ops.bbox3d_12dim_to_8dim(g["b12"])
    assert b8.dtype == g["b8"].dtype and np.array_equal(b8, g["b8"])


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_device_box_formats_equal_reference_python(be):
    """rd_score_filter_dets (10 -> 11 dims + score, every row above the threshold) and rd_dets12_to_8 on the device against the
    reference functions' outputs: bit-equal except the yaw column (device atan2f: 1e-6) and the sqrt / mean columns (1 ulp)."""
    g = np.load(os.path.join(GOLD, "box_formats.npz"))
    import ctypes
    K = g["b10"].shape[0]
    L = be.lib
    sc = np.ascontiguousarray(g["b12"][:, 11])
    wsb = L.raw("rd_score_filter_workspace_bytes")(K)
    ws, out, cnt = be.empty(wsb), be.empty(K * 48), be.empty(4)
    L.call("rd_score_filter_dets", be.ptr(be.up(sc)), be.ptr(be.up(g["b10"])), K, ctypes.c_float(0.0), be.ptr(out), be.ptr(cnt),
           be.ptr(ws), wsb, be.stream)
    assert int(be.down(cnt, np.int32, (1,))[0]) == K
    rows = be.down(out, np.float32, (K, 12))
    assert np.array_equal(rows[:, :8], g["b12"][:, :8]) and np.array_equal(rows[:, 9:], g["b12"][:, 9:])
    assert np.abs(rows[:, 8] - g["b12"][:, 8]).max() < 1e-6
    o8 = be.empty(K * 32)
    L.call("rd_dets12_to_8", be.ptr(be.up(g["b12"])), K, None, be.ptr(o8), be.stream)
    d8 = be.down(o8, np.float32, (K, 8))
    assert np.allclose(d8, g["b8"], rtol=3e-7, atol=1e-6)


def test_assigner_sees_the_state_the_reference_chain_leaves():
    """Bbox3dAssigner after LoadRecord + ProcessMissValue: the point array and mask this package hands to assign3D_v2 are, bit
    for bit, the ones the reference's own stages produce (captured from the reference chain, full-size record)."""
    from rangedet_amd import synth
    from rangedet_amd.core import input as CI
    d = json.load(open(os.path.join(GOLD, "assigner_state_sha256.json")))
    rec = dict(synth.raw_record(1))
    CI.LoadRecord().apply(rec)
    CI.ProcessMissValue().apply(rec)
    pc, mask = CI.Bbox3dAssigner._state_after_earlier_stages(rec)
    assert list(pc.reshape(-1, 3).shape) == d["pc_shape"] and float(mask.sum()) == d["mask_sum"]
    assert hashlib.sha256(np.ascontiguousarray(pc.reshape(-1, 3), np.float32).tobytes()).hexdigest() == d["pc_sha256"]
    assert hashlib.sha256(np.ascontiguousarray(mask.reshape(-1, 1), np.float32).tobytes()).hexdigest() == d["mask_sha256"]
    # without ProcessMissValue in the chain (LoadRecord state only) the arrays differ: the fill is what the pin is about
    rec2 = dict(synth.raw_record(1))
    CI.LoadRecord().apply(rec2)
    pc2, mask2 = CI.Bbox3dAssigner._state_after_earlier_stages(rec2)
    assert mask2.sum() < mask.sum()


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_decode3d_round_trip_with_the_reference_encoder(be):
    """decode_roundtrip.npz: per-point regression targets made by the reference's own GenerateTarget.get_rpn_reg_target
    (rangedet/core/input.py:452-507) for 40 boxes (x, y, z, l, w, h, yaw) and 12 points inside each.  Decode3DBbox
    (operator_cxx/contrib/decode_3d_bbox-inl.h:170-262; it needs MXNet's headers to build, so no compiled reference exists here) is
    that encoder's inverse: the device decode (rd_decode3d_bbox) and the oracle's restatement, followed by the reference's own
    box-format helpers (tools/test.py:43-81, pinned bit for bit by box_formats.npz), must give every box back from every one of its
    points -- which fixes the delta layout [dx, dy, log w, log l, cos, sin, z0, log h], the signed-square offsets, the azimuth frame and
    its rotation direction, and the corner order the 10 -> 11 -> 8 conversion reads the heading and the length / width from."""
    from oracle import cpu_ops as O
    g = np.load(os.path.join(GOLD, "decode_roundtrip.npz"))
    pc, gt, ind, deltas = g["pc"], g["gt"], g["ind"], g["deltas"]
    fg = ind >= 0
    assert fg.sum() > 400 and not np.any(deltas[~fg]) and len(set(ind[fg].tolist())) == len(gt)
    n = len(pc)
    out = be.empty(n * 40)
    be.lib.call("rd_decode3d_bbox", be.ptr(be.up(deltas)), be.ptr(be.up(pc)), be.ptr(out), 1, n, 8, 0, be.stream)
    dev10 = be.down(out, np.float32, (n, 10))
    for name, b10 in (("device", dev10), ("oracle", O.decode3d(deltas[None], pc[None])[0])):
        b11 = O.bbox3d_10dim_to_11dim(b10[fg])
        b8 = O.bbox3d_12dim_to_8dim(np.concatenate([b11, np.ones((b11.shape[0], 1), np.float32)], 1))     # x, y, z, l, w, h, heading, score
        want = gt[ind[fg]]
        err = np.abs(b8[:, :6] - want[:, :6]).max(axis=0)
        dyaw = np.abs((b8[:, 6] - want[:, 6] + np.pi) % (2 * np.pi) - np.pi).max()
        # float32 round trip through sqrt / square, log / exp, cos / sin / atan2 at coordinates of up to 70 m: 1e-4 m, 1e-5 rad
        assert err.max() < 1e-4 and dyaw < 2e-5, (name, err, dyaw)
    assert np.abs(dev10 - O.decode3d(deltas[None], pc[None])[0]).max() < 1e-4
