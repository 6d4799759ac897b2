"""rangedet_amd.load_model: MXNet NDArray-list (.params) reader/writer and the reference's load_checkpoint interface
(utils/load_model.py:5-39).  PARITY UNPINNED: no real checkpoint exists offline; the byte layout is checked against a
hand-assembled file that follows MXNet's NDArray::Save field by field, plus a round trip through the writer."""
import struct

import numpy as np
import pytest

from rangedet_amd import load_model as LM
from rangedet_amd import synth


def test_round_trip_full_weight_set(tmp_path):
    P = synth.make_weights(seed=3, width=64)
    aux = {k for k in P if k.endswith(("_moving_mean", "_moving_var"))}
    blob = {("aux:" if k in aux else "arg:") + k: v for k, v in P.items()}
    f = str(tmp_path / "rangedet-0018.params")
    LM.save(f, blob)
    assert LM.get_latest_ckpt_epoch(str(tmp_path / "rangedet")) == 18
    arg, auxp = LM.load_checkpoint(str(tmp_path / "rangedet"), 18)
    assert set(auxp) == aux and set(arg) == set(P) - aux
    for k, v in P.items():
        got = auxp[k] if k in aux else arg[k]
        assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v)
    Q = LM.load_params(str(tmp_path / "rangedet"), 18)
    assert set(Q) == set(P) and all(np.array_equal(Q[k], P[k]) for k in P)


def test_hand_assembled_bytes_and_dtypes(tmp_path):
    w = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    h = np.array([1.5, -2.0], np.float16)
    i64 = np.array([[7], [8]], np.int64)
    legacy = np.array([9, 10, 11], np.float32)
    b = struct.pack("<QQQ", 0x112, 0, 5)
    b += struct.pack("<Ii", 0xF993FAC9, 0) + struct.pack("<I3q", 3, 2, 3, 4) + struct.pack("<iii", 1, 0, 0) + w.tobytes()   # V2
    b += struct.pack("<I", 0xF993FAC8) + struct.pack("<I1q", 1, 2) + struct.pack("<iii", 2, 3, 2) + h.tobytes()           # V1, saved from gpu(3)
    b += struct.pack("<Ii", 0xF993FAC9, 0) + struct.pack("<I2q", 2, 2, 1) + struct.pack("<iii", 1, 0, 6) + i64.tobytes()
    b += struct.pack("<I1I", 1, 3) + struct.pack("<iii", 1, 0, 0) + legacy.tobytes()                                        # legacy: ndim, uint32 dims
    b += struct.pack("<Ii", 0xF993FAC9, 0) + struct.pack("<I", 0)                                                            # a "none" array
    names = [b"arg:w", b"arg:h", b"aux:i", b"arg:old", b"arg:none"]
    b += struct.pack("<Q", 5) + b"".join(struct.pack("<Q", len(n)) + n for n in names)
    f = tmp_path / "x.params"
    f.write_bytes(b)
    d = LM.load(str(f))
    assert np.array_equal(d["arg:w"], w) and d["arg:w"].dtype == np.float32
    assert np.array_equal(d["arg:h"], h) and d["arg:h"].dtype == np.float16
    assert np.array_equal(d["aux:i"], i64) and np.array_equal(d["arg:old"], legacy) and d["arg:none"] is None
    # a list without names comes back as a list
    f2 = tmp_path / "l.params"
    LM.save(str(f2), [w, i64])
    out = LM.load(str(f2))
    assert isinstance(out, list) and np.array_equal(out[0], w) and np.array_equal(out[1], i64)


def test_v3_numpy_shape_records(tmp_path):
    """V3 records (MXNet built with numpy shape semantics): ndim is a signed int32; -1 = unknown shape = a "none" array,
    ndim 0 = a scalar that carries one element."""
    sc = np.array(2.5, np.float32)
    v = np.arange(6, dtype=np.float32).reshape(2, 3)
    b = struct.pack("<QQQ", 0x112, 0, 3)
    b += struct.pack("<Ii", 0xF993FACA, 0) + struct.pack("<i", 0) + struct.pack("<iii", 1, 0, 0) + sc.tobytes()
    b += struct.pack("<Ii", 0xF993FACA, 0) + struct.pack("<i", -1)
    b += struct.pack("<Ii", 0xF993FACA, 0) + struct.pack("<i2q", 2, 2, 3) + struct.pack("<iii", 1, 0, 0) + v.tobytes()
    b += struct.pack("<Q", 0)
    f = tmp_path / "v3.params"
    f.write_bytes(b)
    out = LM.load(str(f))
    assert out[0].shape == () and out[0] == np.float32(2.5) and out[1] is None and np.array_equal(out[2], v)
    f.write_bytes(struct.pack("<QQQ", 0x112, 0, 1) + struct.pack("<Ii", 0xF993FAC9, 0) + struct.pack("<i", -1))
    with pytest.raises(LM.ParamsFormatError, match="negative ndim"):
        LM.load(str(f))


def test_errors(tmp_path):
    f = tmp_path / "bad.params"
    f.write_bytes(struct.pack("<QQQ", 0x113, 0, 0))
    with pytest.raises(LM.ParamsFormatError, match="not an MXNet"):
        LM.load(str(f))
    good = tmp_path / "g.params"
    LM.save(str(good), {"arg:a": np.ones((4, 4), np.float32)})
    f.write_bytes(good.read_bytes()[:60])
    with pytest.raises(LM.ParamsFormatError, match="truncated"):
        LM.load(str(f))
    b = struct.pack("<QQQ", 0x112, 0, 1) + struct.pack("<Ii", 0xF993FAC9, 1)   # sparse storage
    f.write_bytes(b + b"\0" * 64)
    with pytest.raises(LM.ParamsFormatError, match="sparse"):
        LM.load(str(f))
    with pytest.raises(AssertionError):
        LM.get_latest_ckpt_epoch(str(tmp_path / "nothing"))


def test_evaluate_load_record_reads_the_reference_npz_schema(tmp_path):
    """rangedet_amd.evaluate.load_record == LoadRecord.apply (rangedet/core/input.py:23-38): the npz written by
    datasets/create_range_image_roidb.py:119-124,164 holds 'range_image' / 'pc_vehicle_frame' / 'inclination' / 'azimuth';
    arrays come back as float32."""
    from rangedet_amd import evaluate
    rec = synth.raw_record(3, H=8, W=40)
    f = tmp_path / "1550083467346370.npz"
    np.savez(f, range_image=rec['range_image'].astype(np.float64), pc_vehicle_frame=rec['pc_vehicle_frame'].astype(np.float64),
             inclination=rec['inclination'], azimuth=rec['azimuth'], range_image_mask=np.ones((8, 40)))
    got = evaluate.load_record(dict(pc_url=str(f)))
    assert got['range_image'].dtype == np.float32 and got['pc_vehicle_frame'].dtype == np.float32
    assert np.array_equal(got['range_image'], rec['range_image']) and np.array_equal(got['inclination'], rec['inclination'])
    assert evaluate.load_record(rec) is rec
