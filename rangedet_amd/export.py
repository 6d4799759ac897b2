"""Waymo prediction-bin export without the waymo_open_dataset package: the build's counterpart of the reference's
``tools/create_prediction_bin_3d.py`` (``_create_bbox_prediction`` :26-61, ``_create_pd_file_example`` :64-77, ``main``
:80-104) -- the step after the path.

The reference fills ``metrics_pb2.Object`` / ``label_pb2.Label.Box`` messages and writes ``Objects.SerializeToString()``.
Those generated modules belong to a third-party package that is not in this image, so the proto2 wire format is written
directly.  Schema restated from the public waymo-open-dataset protos (label.proto, protos/metrics.proto):

    Label.Box : center_x=1 center_y=2 center_z=3 width=4 length=5 height=6 heading=7      (all double)
    Label     : box=1 (Box)  metadata=2  type=3 (enum)  id=4 (string)
    Object    : object=1 (Label)  score=2 (float)  overlap_with_nlz=3  context_name=4 (string)
                frame_timestamp_micros=5 (int64)  camera_name=6
    Objects   : objects=1 (repeated Object)

PARITY UNPINNED: the field numbers above are not checkable offline (no waymo_open_dataset, no network); the encoder itself is
checked against google.protobuf on a dynamically built schema with these numbers (tests/test_export.py).  Fields are emitted
in field-number order, as the Python protobuf runtime does, so a matching schema gives byte-identical files.
"""
import pickle as pkl
import struct

type_dict = {'TYPE_UNKNOWN': 0, 'TYPE_VEHICLE': 1, 'TYPE_PEDESTRIAN': 2, 'TYPE_SIGN': 3, 'TYPE_CYCLIST': 4}   # :7-13


def _varint(v):
    v &= (1 << 64) - 1                       # int64 / enum: two's complement, 10 bytes when negative
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _tag(field, wire):
    return _varint((field << 3) | wire)


def _double(field, v):
    return _tag(field, 1) + struct.pack("<d", float(v))


def _bytes(field, b):
    return _tag(field, 2) + _varint(len(b)) + b


def encode_box(center_x, center_y, center_z, length, width, height, heading):
    return (_double(1, center_x) + _double(2, center_y) + _double(3, center_z) + _double(4, width) + _double(5, length) +
            _double(6, height) + _double(7, heading))


def _create_bbox_prediction(bbox3d_xyzlwhyaws, pred_type_id, frame_name, marco_ts):
    """One serialized metrics.Object (create_prediction_bin_3d.py:26-61): box + optional score, empty id, type, context
    name and frame timestamp."""
    b = [float(v) for v in bbox3d_xyzlwhyaws]
    label = _bytes(1, encode_box(b[0], b[1], b[2], b[3], b[4], b[5], b[6]))
    label += _tag(3, 0) + _varint(int(pred_type_id))
    label += _bytes(4, b"")                                   # o.object.id = '' marks the proto2 field present
    o = _bytes(1, label)
    if len(b) == 8:
        o += _tag(2, 5) + struct.pack("<f", b[7])
    o += _bytes(4, str(frame_name).encode("utf-8"))
    o += _tag(5, 0) + _varint(int(marco_ts))
    return o


def serialize_objects(obj_list):
    return b"".join(_bytes(1, o) for o in obj_list)


def _create_pd_file_example(obj_list, filename):
    with open(filename, "wb") as f:
        f.write(serialize_objects(obj_list))


def objects_from_output_dict(output_dict):
    """:86-101: every box of every class of every non-empty frame, in dict order."""
    objs = []
    for rec_id, output in output_dict.items():
        if len(output) == 0:
            continue
        for pred_type, pred_bboxes3d in output['det_xyzlwhyaws'].items():
            for pred_bbox3d in pred_bboxes3d:
                objs.append(_create_bbox_prediction(pred_bbox3d, type_dict[pred_type], output['meta_info']['name'],
                                                    output['meta_info']['timestamp_micros']))
    return objs


def main(pred_boxes_path, config_name, save_bin_dir):
    """:80-104: the pickle tools/test.py:235-237 writes (annotation_dict, then output_dict) -> <save_bin_dir>/<config>.bin"""
    with open(pred_boxes_path, "rb") as fr:
        pkl.load(fr)
        output_dict = pkl.load(fr)
    _create_pd_file_example(objects_from_output_dict(output_dict), '{}/{}.bin'.format(save_bin_dir, config_name))


# ---- reader (tests / inspection) -----------------------------------------------------------------------------------
def _read_varint(buf, o):
    v, s = 0, 0
    while True:
        b = buf[o]
        o += 1
        v |= (b & 0x7F) << s
        s += 7
        if not b & 0x80:
            return v, o


def _fields(buf):
    o = 0
    while o < len(buf):
        key, o = _read_varint(buf, o)
        f, w = key >> 3, key & 7
        if w == 0:
            v, o = _read_varint(buf, o)
        elif w == 1:
            v = struct.unpack_from("<d", buf, o)[0]
            o += 8
        elif w == 5:
            v = struct.unpack_from("<f", buf, o)[0]
            o += 4
        elif w == 2:
            n, o = _read_varint(buf, o)
            v = bytes(buf[o:o + n])
            o += n
        else:
            raise ValueError("wire type %d" % w)
        yield f, w, v


def parse_objects(data):
    """Serialized metrics.Objects -> list of dicts (box fields, type, score, context_name, frame_timestamp_micros)."""
    out = []
    names = {1: "center_x", 2: "center_y", 3: "center_z", 4: "width", 5: "length", 6: "height", 7: "heading"}
    for f, w, v in _fields(data):
        if f != 1:
            continue
        d = {}
        for f2, _, v2 in _fields(v):
            if f2 == 1:
                for f3, _, v3 in _fields(v2):
                    if f3 == 1:
                        d.update({names[f4]: v4 for f4, _, v4 in _fields(v3)})
                    elif f3 == 3:
                        d["type"] = v3
                    elif f3 == 4:
                        d["id"] = v3.decode("utf-8")
            elif f2 == 2:
                d["score"] = v2
            elif f2 == 4:
                d["context_name"] = v2.decode("utf-8")
            elif f2 == 5:
                d["frame_timestamp_micros"] = v2 - (1 << 64) if v2 >= 1 << 63 else v2
        out.append(d)
    return out
