"""ORACLE (test infrastructure only -- never imported by rangedet_amd/): numpy restatement of the reference's test-time
input transform chain, the checker of rd_input_transform (csrc/k_input.h).

  LoadRecord            rangedet/core/input.py:14-42        mask = range > 0, points of masked pixels zeroed
  ProcessMissValue      rangedet/core/input.py:89-137       -1 returns filled from the right neighbour, then [80,0,0,-1];
                                                            "car window" pixels zeroed
  SepAndClipData        rangedet/core/input.py:140-160      per-channel clip (azimuth popped, :149)
  GetUnnormalizedRange  rangedet/core/input.py:172-180
  NormData              rangedet/core/input.py:183-198      (x - mean) / sqrt(var)
  GetCoordinates        rangedet/core/input.py:201-214
  CombineData           rangedet/core/input.py:217-229      8 channels in the config's order
  PadData / TransposeData   rangedet/core/input.py:522-557  zero pad on the right, CHW
  GenerateFPNTarget / TransAndReshape   rangedet/core/input.py:560-624
  constants             config/rangedet/rangedet_veh_wo_aug_4_18e.py:245-282, :71

PARITY PINNED (round 3): bit-equal to the outputs of the reference's own transform objects (rangedet/core/input.py, with the
parameter classes of the reference config) on the committed fixtures tests/golden/input_chain_*.npz and, through SHA-256
digests, on two full-size 64x2650 records (tests/golden/make_ref_python_golden.py ran the reference chain in the build
container with inert stand-ins for its unused module-scope imports; tests/test_ref_python_pins.py).
The constants are restated here on purpose (not imported from the product): tests compare them with the product's.
"""
import numpy as np

CLIP = {  # config:245-255 (azimuth popped, input.py:149)
    'range_value': (0, 80), 'intensity': (0, 1), 'elongation': (0, 1),
    'pc_vehicle_frame_x': (-80, 80), 'pc_vehicle_frame_y': (-80, 80), 'pc_vehicle_frame_z': (-5, 10),
    'inclination': (-0.5, 0.1),
}
NORM = {  # config:257-267 (mean, var)
    'range_value': (20.0, 1500.0), 'intensity': (0.1, 0.01), 'elongation': (7.2558375e-02, 2.6764875e-02),
    'pc_vehicle_frame_x': (1.5672500e+00, 3.0740625e+02), 'pc_vehicle_frame_y': (9.8824875e-01, 2.1913250e+02),
    'pc_vehicle_frame_z': (1.4, 1.0), 'inclination': (-8.8427375e-02, 9.9001750e-03),
    'azimuth': (-7.8061250e-03, 2.5494125e+00),
}
COMBINE = ['range_value', 'intensity', 'elongation', 'pc_vehicle_frame_x', 'pc_vehicle_frame_y',
           'pc_vehicle_frame_z', 'inclination', 'azimuth']  # config:269-282
INTERVAL = {1: (30, 100), 2: (15, 30), 4: (0, 15)}  # config:71
FPN_STRIDES = (1, 2, 4)


def _fill_noise(data, miss, width):
    shifted = data[:, list(range(1, width)) + [0], :]
    data[miss, :] = shifted[miss, :]
    return data


def transform(rec, pad_hw=(64, 2656)):
    """Test-mode transform chain -> the named float32 arrays one frame feeds to the graph (batch dim added)."""
    ri = rec['range_image'].astype(np.float32).copy()
    pc = rec['pc_vehicle_frame'].astype(np.float32).copy()
    mask = ri[..., 0:1] > 0                                   # LoadRecord (input.py:40-42)
    pc[~mask[..., 0]] = 0
    H, W, _ = ri.shape
    # ProcessMissValue (input.py:105-137)
    rmask = (ri[..., 0] > 0)
    miss = ri[:, :, 0] == -1
    ri = _fill_noise(ri, miss, W)
    pc = _fill_noise(pc, miss, W)
    rmask = _fill_noise(rmask[:, :, None].copy(), miss, W).squeeze()
    still = ri[:, :, 0] == -1
    r0 = ri[:, :, 0]
    dn = r0[[H - 2, H - 1] + list(range(H - 2)), :]
    up = r0[list(range(2, H)) + [0, 1], :]
    rt = r0[:, [W - 2, W - 1] + list(range(W - 2))]
    lf = r0[:, list(range(2, W)) + [0, 1]]
    car = still & ((dn != -1) | (up != -1) | (rt != -1) | (lf != -1))
    ri[still, :] = np.array([80, 0, 0, -1], np.float32)
    pc[still, :] = 0
    ri[car, :] = np.array([0, 0, 0, -1], np.float32)
    pc[car, :] = 0
    rmask = rmask.astype(np.float32)[:, :, None]
    # SepAndClipData / GetUnnormalizedRange / NormData / GetCoordinates / CombineData
    f = {
        'range_value': ri[:, :, 0].copy(), 'intensity': ri[:, :, 1].copy(), 'elongation': ri[:, :, 2].copy(),
        'pc_vehicle_frame_x': pc[:, :, 0].copy(), 'pc_vehicle_frame_y': pc[:, :, 1].copy(),
        'pc_vehicle_frame_z': pc[:, :, 2].copy(),
        'inclination': np.tile(rec['inclination'].astype(np.float32)[:, None], (1, W)),
    }
    f['azimuth'] = np.arctan2(f['pc_vehicle_frame_y'], f['pc_vehicle_frame_x'])
    for n, (lo, hi) in CLIP.items():
        f[n] = np.clip(f[n], lo, hi)
    unnorm = f['range_value'][:, :, None].copy()
    for n, (mean, var) in NORM.items():
        f[n] = (f[n] - mean) / (var ** 0.5)
    coord = np.stack([f['pc_vehicle_frame_x'], f['pc_vehicle_frame_y'], f['pc_vehicle_frame_z']], 2)
    data = np.stack([f[n] for n in COMBINE], 2)

    def pad(a):                                                # PadData (input.py:539-544)
        out = np.zeros((pad_hw[0], pad_hw[1], a.shape[-1]), np.float32)
        out[:a.shape[0], :a.shape[1]] = a
        return out

    data, rmask, pcp, unnorm, coord = (pad(a).transpose(2, 0, 1) for a in (data, rmask, pc, unnorm, coord))
    out = {'input_data': data[None], 'coord_s1': coord[None]}
    for s in FPN_STRIDES:                                      # GenerateFPNTarget + TransAndReshape
        lo, hi = INTERVAL[s]
        m = ((lo <= unnorm) & (unnorm < hi)).astype(np.float32)
        sl = slice(s // 2, None, s)
        out['range_image_mask_s%d' % s] = (rmask * m)[:, :, sl].reshape(-1)[None].astype(np.float32)
        out['pc_vehicle_frame_s%d' % s] = pcp[:, :, sl].reshape(3, -1).transpose(1, 0)[None].astype(np.float32).copy()
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in out.items()}


def make_frame(idx, W=2650, pad_W=2656, H=64):
    """Synthetic record idx (the product's generator of RAW records) through the restated chain: numpy arrays with batch dim."""
    from rangedet_amd import synth
    return transform(synth.raw_record(idx, H, W), (H, pad_W))


def make_batch(idxs, W=2650, pad_W=2656, H=64):
    frames = [make_frame(i, W, pad_W, H) for i in idxs]
    return {k: np.concatenate([f[k] for f in frames], 0) for k in frames[0]}
