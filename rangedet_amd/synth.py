"""Synthetic inputs and weights for the RangeDet hot path (no dataset or checkpoint is available offline).

``raw_record`` draws a record with the npz schema of datasets/create_range_image_roidb.py:119-124,164; ``make_batch`` runs
such records through the device transform chain (rd_input_transform: LoadRecord ... TransAndReshape,
rangedet/core/input.py:14-42,89-229,522-624, constants config/rangedet/rangedet_veh_wo_aug_4_18e.py:245-282) and returns the
tensors under the names the graph consumes (config:400-404): input_data, coord_s1, pc_vehicle_frame_s{1,2,4},
range_image_mask_s{1,2,4}.  ``make_weights`` draws seeded parameters under the reference's MXNet names.
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


def raw_record(idx, H=64, W=2650):
    """A synthetic npz record (schema of datasets/create_range_image_roidb.py:119-124,164)."""
    rng = np.random.default_rng(2650 + idx)
    incl = np.linspace(0.04, -0.31, H).astype(np.float32)
    az = (((np.arange(W, 0, -1) - 0.5) / W * 2 - 1) * np.pi).astype(np.float32)
    r = rng.uniform(1, 75, (H, W))
    k = np.ones(5) / 5
    r = np.apply_along_axis(lambda v: np.convolve(np.pad(v, 2, mode='wrap'), k, mode='valid'), 1, r)
    miss = np.zeros((H, W), bool)
    n_runs = int(0.12 * H * W / 6)
    hs = rng.integers(0, H, n_runs)
    ws = rng.integers(0, W, n_runs)
    ln = rng.integers(1, 12, n_runs)
    for h, w, l in zip(hs, ws, ln):
        miss[h, w:w + l] = True
    inten = rng.uniform(0, 1, (H, W))
    elong = rng.uniform(0, 0.3, (H, W))
    x = r * np.cos(incl)[:, None] * np.cos(az)[None, :]
    y = r * np.cos(incl)[:, None] * np.sin(az)[None, :]
    z = r * np.sin(incl)[:, None] + 2.0
    ri = np.stack([r, inten, elong, np.zeros_like(r)], 2).astype(np.float32)
    pc = np.stack([x, y, z], 2).astype(np.float32)
    ri[miss] = -1
    pc[miss] = 0
    return dict(range_image=ri, pc_vehicle_frame=pc, inclination=incl, azimuth=az)


def make_batch(idxs, W=2650, pad_W=2656, H=64, lib=None, alloc=None):
    """Synthetic records `idxs` through the DEVICE transform chain (rd_input_transform): the named float32 device tensors
    (with batch dim) the graph consumes.  Needs the GPU -- the numpy restatement of the chain is test infrastructure and lives
    in oracle/input_ref.py."""
    from .input_transform import DeviceInputTransform
    return DeviceInputTransform(pad_hw=(H, pad_W), lib=lib, alloc=alloc)([raw_record(i, H, W) for i in idxs])


def make_frame(idx, W=2650, pad_W=2656, H=64, lib=None, alloc=None):
    return make_batch([idx], W, pad_W, H, lib, alloc)


# ---- weights --------------------------------------------------------------------------------------------------
NUM_BLOCK = {'res1': 2, 'res2a': 3, 'res2': 3, 'res3a': 5, 'res3': 5, 'agg1': 2, 'agg2': 2, 'agg2a': 1, 'agg3': 2}
NUM_FILTER = {'res1': 64, 'res2a': 64, 'res2': 128, 'res3a': 128, 'res3': 128, 'agg1': 64, 'agg2': 128, 'agg2a': 64, 'agg3': 64}
# reg_delta layout [dx, dy, log w, log l, cos yaw, sin yaw, z0, log h] (decode_3d_bbox-inl.h:186-193): car-sized boxes
REG_BIAS = np.array([0.0, 0.0, 0.7, 1.5, 1.0, 0.0, -1.0, 0.5], np.float32)
GAIN = 1.0            # conv weight variance = GAIN / fan_in (keeps activations O(1-10) through the 60+ layers)
GAMMA = (0.5, 1.0)    # BN gamma range
CLS_STD, REG_STD = 0.5, 0.05  # head output weights: logits with O(0.5) spread, deltas = car-sized bias + small noise
CLS_BIAS = -1.807  # calibrated with the oracle (seed 18, frame 0): ~1500 of the valid pixels score > 0.5


def make_weights(seed=18, width=2656, in_ch=8, cls_bias=CLS_BIAS, num_classes=1, num_filter=None):
    """num_filter: {stage: channels} overrides of NUM_FILTER (BackboneParam.num_filter of the reference config); the widths must be
    consistent the way the reference's graph needs them (agg2 = res2, agg1 = res1, agg2a = res2a, agg3 = agg1: the stages a skip
    connection adds, dla_backbone.py:117-127,143-150; res1 = 64: the Meta-Kernel unit's data channels)."""
    rng = np.random.default_rng(seed)
    P = {}
    NF = dict(NUM_FILTER, **(num_filter or {}))

    def conv_w(name, o, i, kh, kw, bias=False, std=None):
        s = std if std is not None else np.sqrt(GAIN / (i * kh * kw))
        P[name + "_weight"] = rng.normal(0, s, (o, i, kh, kw)).astype(np.float32)
        if bias:
            P[name + "_bias"] = np.zeros(o, np.float32)

    def bn_p(name, c):
        P[name + "_gamma"] = rng.uniform(GAMMA[0], GAMMA[1], c).astype(np.float32)
        P[name + "_beta"] = rng.normal(0, 0.1, c).astype(np.float32)
        P[name + "_moving_mean"] = rng.normal(0, 0.1, c).astype(np.float32)
        P[name + "_moving_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)

    def block(name, cin, f, proj, meta=False):
        if meta:
            pre = name + "_%d" % width
            conv_w(pre + "_mlp0", 32, 3, 1, 1, bias=True)
            P[pre + "_mlp0_bias"] = rng.normal(0, 0.1, 32).astype(np.float32)
            conv_w(pre + "_mlp1", 64, 32, 1, 1, bias=True)
            P[pre + "_mlp1_bias"] = rng.normal(0, 0.1, 64).astype(np.float32)
            bn_p(name + "point_wise_mlp_bn1", 576)
            conv_w(name + "aggregation_conv1", f, 576, 1, 1)
            bn_p(name + "aggregation_bn1", f)
        else:
            conv_w(name + "_conv1", f, cin, 3, 3)
            bn_p(name + "_bn1", f)
        conv_w(name + "_conv2", f, f, 3, 3)
        bn_p(name + "_bn2", f)
        if proj:
            conv_w(name + "_sc", f, cin, 1, 1)
            bn_p(name + "_sc_bn", f)

    def stage(name, cin, f, nblk, meta_units=()):
        block(name + "_unit1", cin, f, True, (name + "_unit1") in meta_units)
        for i in range(2, nblk + 1):
            block("%s_unit%d" % (name, i), f, f, False, ("%s_unit%d" % (name, i)) in meta_units)

    stage('res1', in_ch, NF['res1'], NUM_BLOCK['res1'], meta_units=('res1_unit2',))
    stage('res2a', NF['res1'], NF['res2a'], NUM_BLOCK['res2a'])
    stage('res2', NF['res2a'], NF['res2'], NUM_BLOCK['res2'])
    stage('res3a', NF['res2'], NF['res3a'], NUM_BLOCK['res3a'])
    stage('res3', NF['res3a'], NF['res3'], NUM_BLOCK['res3'])
    # (deconv input = the upsampled stage: res3 -> agg2, res2 -> agg1, agg2 -> agg2a, agg2a -> agg3; dla_backbone.py:143-150)
    for name, cin, f, k in (("agg2", NF['res3'], NF['agg2'], (3, 8)), ("agg1", NF['res2'], NF['agg1'], (3, 8)),
                            ("agg2a", NF['agg2'], NF['agg2a'], (3, 4)), ("agg3", NF['agg2a'], NF['agg3'], (3, 4))):
        s = np.sqrt(GAIN / (cin * k[0] * k[1] / (k[1] // 2)))
        P[name + "_deconv_weight"] = rng.normal(0, s, (cin, f, k[0], k[1])).astype(np.float32)  # (I,O,kh,kw)
        bn_p(name + "_deconv_bn", f)
        stage(name + "_res", f, f, NUM_BLOCK[name])
    lvl_in = {0: NF['agg3'] + in_ch, 1: NF['agg2a'], 2: NF['agg2']}
    for lvl in range(3):
        for tower in ("cls", "reg"):
            cin = lvl_in[lvl]
            for i in range(4):
                n = 'rpn_%s_conv_%d_lvl_%d' % (tower, i, lvl)
                conv_w(n, 128, cin, 3, 3)
                bn_p(n + "_bn", 128)
                cin = 128
        conv_w('rpn_cls_logit_lvl_%d' % lvl, num_classes, 128, 1, 1, bias=True, std=CLS_STD)
        P['rpn_cls_logit_lvl_%d_bias' % lvl] = np.full(num_classes, cls_bias, np.float32)
        conv_w('rpn_reg_delta_lvl_%d' % lvl, 8 * num_classes, 128, 1, 1, bias=True, std=REG_STD)
        P['rpn_reg_delta_lvl_%d_bias' % lvl] = np.tile(REG_BIAS, num_classes)
    return P


def cluster_dets(n_obj, rep, seed=7, spread=60.0, jitter=0.05, quant=None):
    """Clustered (K,12) detections for the WNMS micro-benchmark / tests."""
    rng = np.random.default_rng(seed)
    ctr = rng.uniform(-spread, spread, (n_obj, 2))
    yaw = rng.uniform(-np.pi, np.pi, n_obj)
    l = rng.uniform(3.5, 5.5, n_obj)
    w = rng.uniform(1.6, 2.2, n_obj)
    rows = []
    for o in range(n_obj):
        for _ in range(rep):
            c = ctr[o] + rng.normal(0, jitter, 2)
            y = yaw[o] + rng.normal(0, 0.02)
            if rng.uniform() < 0.05:
                y += np.pi
            L = l[o] + rng.normal(0, 0.05)
            Wd = w[o] + rng.normal(0, 0.03)
            cs, sn = np.cos(y), np.sin(y)
            cor = np.array([[L / 2, -Wd / 2], [-L / 2, -Wd / 2], [-L / 2, Wd / 2], [L / 2, Wd / 2]])
            pts = cor @ np.array([[cs, -sn], [sn, cs]]).T + c
            z0 = rng.uniform(-1, 0.5)
            h = rng.uniform(1.4, 2.0)
            yawc = np.arctan2(pts[0, 1] - pts[1, 1], pts[0, 0] - pts[1, 0])
            rows.append(list(pts.reshape(-1)) + [yawc, z0, h, rng.uniform(0.5, 1.0)])
    d = np.array(rows, dtype=np.float32)
    d = d[rng.permutation(len(d))]
    if quant:
        d[:, 11] = np.round(d[:, 11] * quant) / quant
    return d
