"""Checkpoint loading without MXNet: the build's counterpart of the reference's ``utils/load_model.py``
(``get_latest_ckpt_epoch`` :5-15, ``load_checkpoint`` :18-39) -- same names, same (arg_params, aux_params) result,
numpy arrays instead of ``mx.nd.NDArray``.

``mx.nd.load`` reads MXNet's NDArray-list binary.  MXNet is a third-party dependency that is not vendored in
/root/reference; the format below is restated from its public serialization code (src/ndarray/ndarray.cc
``NDArray::Save`` / ``NDArray::Load`` and ``MXNDArrayLoad``, MXNet 1.x):

    uint64  0x112                      list magic (kMXAPINDArrayListMagic)
    uint64  0                          reserved
    uint64  N                          number of arrays
    N x NDArray:
        uint32  magic                  0xF993fac9 = V2, 0xF993faca = V3 (numpy shape semantics), 0xF993fac8 = V1;
                                       anything else = legacy format: the word is ndim, dims are uint32
        int32   storage type           V2/V3 only; 0 = dense (the only kind a checkpoint of this model holds)
        int32   ndim ; int64 dim[ndim] V1/V2/V3 shape (legacy: uint32 dims)
        (a "none" array ends here: ndim 0 for V1/V2/legacy, ndim -1 for V3 whose ndim 0 is a scalar)
        int32 dev_type ; int32 dev_id  context it was saved from
        int32   type flag              0 f32, 1 f64, 2 f16, 3 u8, 4 i32, 5 i8, 6 i64
        raw little-endian data         prod(shape) * itemsize bytes
    uint64  M                          number of names (0 or N)
    M x { uint64 length ; bytes }      "arg:<name>" / "aux:<name>"

PARITY UNPINNED: no real ``.params`` file is available offline, so the reader is validated only against the writer in
this module (round trip) and against the byte layout above; validate against a real checkpoint before relying on it.
"""
import glob
import struct

import numpy as np

LIST_MAGIC = 0x112
V1_MAGIC, V2_MAGIC, V3_MAGIC = 0xF993FAC8, 0xF993FAC9, 0xF993FACA
DTYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
FLAGS = {np.dtype(v): k for k, v in DTYPES.items()}


class ParamsFormatError(ValueError):
    pass


class _Reader:
    def __init__(self, buf):
        self.b, self.o = memoryview(buf), 0

    def take(self, fmt):
        n = struct.calcsize(fmt)
        if self.o + n > len(self.b):
            raise ParamsFormatError("truncated file at byte %d" % self.o)
        v = struct.unpack_from(fmt, self.b, self.o)
        self.o += n
        return v if len(v) > 1 else v[0]

    def raw(self, n):
        if self.o + n > len(self.b):
            raise ParamsFormatError("truncated array data at byte %d (need %d more bytes)" % (self.o, n))
        v = self.b[self.o:self.o + n]
        self.o += n
        return v


def _read_ndarray(r):
    magic = r.take("<I")
    if magic in (V2_MAGIC, V3_MAGIC):
        stype = r.take("<i")
        if stype != 0:
            raise ParamsFormatError("sparse storage type %d is not supported (dense checkpoints only)" % stype)
        ndim = r.take("<i")                                # int32: V3 (numpy shape semantics) stores -1 for "unknown"
        if ndim < 0:
            if magic == V3_MAGIC and ndim == -1:
                return None                                # "none" array under numpy semantics: nothing else is stored
            raise ParamsFormatError("negative ndim %d" % ndim)
        shape = tuple(r.take("<%dq" % ndim)) if ndim > 1 else ((r.take("<q"),) if ndim == 1 else ())
    elif magic == V1_MAGIC:
        ndim = r.take("<I")
        shape = tuple(r.take("<%dq" % ndim)) if ndim > 1 else ((r.take("<q"),) if ndim == 1 else ())
    else:  # legacy: the word just read is ndim, uint32 dims
        ndim = magic
        if ndim > 32:
            raise ParamsFormatError("not an NDArray record (magic 0x%08x)" % magic)
        shape = tuple(r.take("<%dI" % ndim)) if ndim > 1 else ((r.take("<I"),) if ndim == 1 else ())
    if ndim == 0 and magic != V3_MAGIC:
        return None                                        # "none" array: nothing else is stored
    if any(d < 0 for d in shape):
        raise ParamsFormatError("negative dimension in shape %r" % (shape,))
    r.take("<ii")                                          # context (dev_type, dev_id)
    flag = r.take("<i")
    if flag not in DTYPES:
        raise ParamsFormatError("unknown type flag %d" % flag)
    dt = np.dtype(DTYPES[flag]).newbyteorder("<")
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    return np.frombuffer(r.raw(n * dt.itemsize), dtype=dt, count=n).reshape(shape).copy()


def load(fname):
    """mx.nd.load: dict name -> array when the file carries names, else a list of arrays."""
    with open(fname, "rb") as f:
        r = _Reader(f.read())
    magic, _reserved = r.take("<QQ")
    if magic != LIST_MAGIC:
        raise ParamsFormatError("%s: not an MXNet NDArray list (magic 0x%x)" % (fname, magic))
    arrays = [_read_ndarray(r) for _ in range(r.take("<Q"))]
    nnames = r.take("<Q")
    if nnames == 0:
        return arrays
    if nnames != len(arrays):
        raise ParamsFormatError("%d names for %d arrays" % (nnames, len(arrays)))
    names = [bytes(r.raw(r.take("<Q"))).decode("utf-8") for _ in range(nnames)]
    return dict(zip(names, arrays))


def save(fname, data):
    """mx.nd.save for dense arrays (V2 records): dict name -> array, or a list of arrays."""
    names = list(data.keys()) if isinstance(data, dict) else []
    arrays = [np.asarray(data[k]) for k in names] if names else [np.asarray(a) for a in data]
    out = [struct.pack("<QQQ", LIST_MAGIC, 0, len(arrays))]
    for a in arrays:
        if a.dtype not in FLAGS:
            raise ParamsFormatError("dtype %s has no MXNet type flag" % a.dtype)
        if a.ndim == 0:
            a = a.reshape(1)
        out.append(struct.pack("<Ii", V2_MAGIC, 0))
        out.append(struct.pack("<I%dq" % a.ndim, a.ndim, *a.shape))
        out.append(struct.pack("<iii", 1, 0, FLAGS[a.dtype]))
        out.append(np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<"), copy=False).tobytes())
    out.append(struct.pack("<Q", len(names)))
    for k in names:
        kb = k.encode("utf-8")
        out.append(struct.pack("<Q", len(kb)) + kb)
    with open(fname, "wb") as f:
        f.write(b"".join(out))


# ---- the reference's interface (utils/load_model.py) ----------------------------------------------------------------
def get_latest_ckpt_epoch(prefix):
    """utils/load_model.py:5-15."""
    def get_checkpoint_epoch(p):
        return int(p[p.rfind('.params') - 4:p.rfind('.params')])

    checkpoints = glob.glob(prefix + '*.params')
    assert len(checkpoints), 'can not find params startswith {}'.format(prefix)
    return max(get_checkpoint_epoch(x) for x in checkpoints)


def load_checkpoint(prefix, epoch):
    """utils/load_model.py:18-39: (arg_params, aux_params), numpy arrays keyed by parameter name."""
    save_dict = load('%s-%04d.params' % (prefix, epoch))
    if not isinstance(save_dict, dict):
        raise ParamsFormatError("checkpoint without parameter names")
    arg_params, aux_params = {}, {}
    for k, v in save_dict.items():
        tp, name = k.split(':', 1)
        if tp == 'arg':
            arg_params[name] = v
        if tp == 'aux':
            aux_params[name] = v
    return arg_params, aux_params


def load_params(prefix, epoch):
    """One dict name -> float32 array with both argument and auxiliary (BatchNorm moving statistics) states: the form
    ``runtime.Executor`` / ``RangeDetPipeline`` take."""
    arg, aux = load_checkpoint(prefix, epoch)
    P = {k: np.asarray(v, np.float32) for k, v in arg.items()}
    P.update({k: np.asarray(v, np.float32) for k, v in aux.items()})
    return P
