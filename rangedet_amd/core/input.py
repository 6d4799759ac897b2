"""``rangedet.core.input`` on the HIP path: the sixteen transform classes the reference config imports
(config/rangedet/rangedet_veh_wo_aug_4_18e.py:15-21), with the reference's constructor parameters, so that the config's
``transform = [LoadRecord(), LoadGTInfo(), FilterGTClass(...), ProcessMissValue(), SepAndClipData(ClipDataParam), ...,
TransAndReshape(TransAndReshapeParam)]`` list (config:380-399) builds and runs unchanged.

Design (not the reference's): the image-sized stages do not touch pixels on the host.  LoadRecord reads the raw arrays;
every stage from ProcessMissValue to GenerateFPNTarget only RECORDS its parameters in the record's chain description;
TransAndReshape -- the last stage of the test chain -- validates that description and launches ONE fused kernel
(rd_input_transform, csrc/k_input.h) that writes the named float32 tensors the graph consumes (input_data, coord_s1,
pc_vehicle_frame_s{1,2,4}, range_image_mask_s{1,2,4}) on the device.  ``run_chain(transform, records)`` does the same for a
whole batch of records with one launch.  The per-object stages (LoadGTInfo, FilterGTClass, GetFixedLengthGTBbox) are small
host-side array edits and run as such; Bbox3dAssigner runs on the GPU (rd_assign3d_v2); GenerateTarget (training targets)
is outside the inference path and raises.

Semantics of every stage: rangedet/core/input.py (cited per class); checked against the numpy restatement in
oracle/input_ref.py by tests/test_kernels.py::test_input_transform and tests/test_core_input.py.
"""
import numpy as np

EPS = 1e-3            # rangedet/core/input.py:10
_CHAIN = "_rd_chain"  # key of the recorded chain description inside an input_record


class DetectionAugmentation:
    def apply(self, input_record):
        raise NotImplementedError


def _stage(rec, name, **params):
    rec.setdefault(_CHAIN, []).append((name, params))


class LoadRecord(DetectionAugmentation):
    """rangedet/core/input.py:14-42: raw arrays of the frame's npz (`pc_url`), float32.  The validity mask (range > 0) and the
    zeroing of invalid points happen inside the device kernel."""

    def apply(self, input_record):
        if 'range_image' not in input_record:
            with np.load(input_record["pc_url"]) as z:
                for k in ('pc_vehicle_frame', 'range_image', 'inclination', 'azimuth'):
                    input_record[k] = z[k].astype(np.float32)
        else:
            for k in ('pc_vehicle_frame', 'range_image', 'inclination'):
                input_record[k] = np.asarray(input_record[k], np.float32)
        input_record[_CHAIN] = [("LoadRecord", {})]


class LoadGTInfo(DetectionAugmentation):
    """rangedet/core/input.py:45-59: ground-truth arrays to float32 (absent keys are left absent: test-time records of the
    unlabeled split carry none)."""
    KEYS = ("gt_class", "gt_bbox_yaw", "gt_bbox_csa", "gt_bbox_imu", "meta_data", "points_in_box")

    def apply(self, input_record):
        for k in self.KEYS:
            if k in input_record:
                input_record[k] = np.asarray(input_record[k]).astype(np.float32)


class FilterGTClass(DetectionAugmentation):
    """rangedet/core/input.py:62-86: keep the ground truth of the trained classes; an empty set becomes one all-zero box."""
    KEYS = ("gt_bbox_imu", "gt_bbox_csa", "gt_bbox_yaw", "points_in_box")
    EMPTY = {"gt_class": (1,), "gt_bbox_imu": (1, 8, 3), "gt_bbox_csa": (1, 7), "gt_bbox_yaw": (1,), "points_in_box": (1,)}

    def __init__(self, valid_class):
        self.valid_class = valid_class

    def apply(self, input_record):
        if "gt_class" not in input_record:
            return
        cls = input_record["gt_class"]
        if cls.size > 0:
            keep = np.isin(cls, list(self.valid_class))
            input_record["gt_class"] = cls[keep]
            for k in self.KEYS:
                if k in input_record:
                    input_record[k] = input_record[k][keep]
        if input_record["gt_class"].size == 0:
            for k, shp in self.EMPTY.items():
                input_record[k] = np.zeros(shp, np.float32)


class ProcessMissValue(DetectionAugmentation):
    """rangedet/core/input.py:89-137 (fill from the right neighbour, then [80,0,0,-1]; car-window pixels zeroed)."""

    def __init__(self):
        self.pc_fill_value = np.array([0, 0, 0])
        self.range_fill_value = np.array([80, 0, 0, -1])

    def apply(self, input_record):
        _stage(input_record, "ProcessMissValue", range_fill=tuple(self.range_fill_value), pc_fill=tuple(self.pc_fill_value))


class SepAndClipData(DetectionAugmentation):
    """rangedet/core/input.py:140-171; azimuth is removed from the clip table (:149) -- on a copy, the param class is not edited."""

    def __init__(self, param):
        self.clip_data_dict = dict(param.clip_data_dict)
        self.clip_data_dict.pop('azimuth', None)

    def apply(self, input_record):
        _stage(input_record, "SepAndClipData", clip=dict(self.clip_data_dict))


class GetUnnormalizedRange(DetectionAugmentation):
    """rangedet/core/input.py:174-183."""

    def apply(self, input_record):
        _stage(input_record, "GetUnnormalizedRange")


class NormData(DetectionAugmentation):
    """rangedet/core/input.py:186-197: (x - mean) / sqrt(var)."""

    def __init__(self, param):
        self.norm_data_dict = param.norm_data_dict

    def apply(self, input_record):
        _stage(input_record, "NormData", norm=dict(self.norm_data_dict))


class GetCoordinates(DetectionAugmentation):
    """rangedet/core/input.py:200-213."""

    def apply(self, input_record):
        _stage(input_record, "GetCoordinates")


class CombineData(DetectionAugmentation):
    """rangedet/core/input.py:216-229."""

    def __init__(self, param):
        self.combine_name_dict = param.combine_name_dict

    def apply(self, input_record):
        _stage(input_record, "CombineData", combine={k: list(v) for k, v in self.combine_name_dict.items()})


class GetFixedLengthGTBbox(DetectionAugmentation):
    """rangedet/core/input.py:232-273: per class the BEV corners (first 4 corners, x/y) of its boxes in a (fixed_length, 8)
    array; unused rows hold the degenerate box [0,0,0,EPS,EPS,EPS,EPS,0] that rotated IoU maps to 0."""
    TYPES = {'TYPE_UNKNOWN': 0, 'TYPE_VEHICLE': 1, 'TYPE_PEDESTRIAN': 2, 'TYPE_SIGN': 3, 'TYPE_CYCLIST': 4}

    def __init__(self, param):
        self.class_type = param.class_type
        self.fixed_length = param.fixed_length

    def apply(self, input_record):
        for c_type in self.class_type:
            name = '_'.join(['gt_bbox', c_type[5:8].lower(), 'for_iou_pred'])
            input_record[name] = self.get_fixed_length_gt_bbox(input_record['gt_bbox_imu'], input_record['gt_class'], c_type,
                                                               self.fixed_length)

    @classmethod
    def get_fixed_length_gt_bbox(cls, gt_bbox, gt_class, class_type, fixed_length=200):
        if gt_bbox.shape[0] != gt_class.shape[0] or gt_bbox.shape[1:] != (8, 3):
            raise ValueError("gt_bbox %s / gt_class %s" % (gt_bbox.shape, gt_class.shape))
        out = np.tile(np.array([0, 0, 0, EPS, EPS, EPS, EPS, 0], np.float32), (fixed_length, 1))
        sel = gt_bbox[gt_class == cls.TYPES[class_type]][:, :4, :2].reshape(-1, 8)
        if sel.shape[0] >= fixed_length:
            raise ValueError("The number of GT boxes is greater than %d" % fixed_length)
        out[:sel.shape[0]] = sel
        return out


class Bbox3dAssigner(DetectionAugmentation):
    """rangedet/core/input.py:276-320 on the GPU (processing_cxx.assign3D_v2 -> rd_assign3d_v2): the index of the ground-truth
    box every point lies in.  The reference reads `pc_vehicle_frame` and `range_image_mask` AS THE EARLIER STAGES LEFT THEM: after
    LoadRecord (mask = range > 0, masked points zeroed, :40-42) and, in the training chain where this stage runs
    (config:345-366), after ProcessMissValue -- missing returns take their right neighbour's point and mask, what is still
    missing becomes a zero point (:105-137).  The image stages of this package only record themselves, so this stage applies
    exactly those two state changes (a few numpy index operations) to its own copies when the chain recorded ProcessMissValue."""

    def __init__(self, param=None):
        self.height, self.width = param.feat_size[0], param.feat_size[1]

    @staticmethod
    def _state_after_earlier_stages(input_record):
        ri = np.asarray(input_record['range_image'], np.float32)
        pc = np.asarray(input_record['pc_vehicle_frame'], np.float32).copy()
        mask = ri[..., 0] > 0                                            # LoadRecord, input.py:40-42
        pc[~mask] = 0
        if any(name == "ProcessMissValue" for name, _ in input_record.get(_CHAIN, [])):
            W = ri.shape[1]
            miss = ri[:, :, 0] == -1                                     # input.py:112
            nb = list(range(1, W)) + [0]                                 # fill_noise: the right neighbour, wrapping (:99-103)
            r0 = ri[:, :, 0].copy()
            r0[miss] = ri[:, nb, 0][miss]
            pc[miss] = pc[:, nb][miss]
            mask = mask.copy()
            mask[miss] = mask[:, nb][miss]
            pc[r0 == -1] = 0                                             # still missing (far fill or car window): zero point (:128-135)
        return pc, mask.astype(np.float32)

    def apply(self, input_record):
        from .. import processing_cxx
        gt = np.asarray(input_record['gt_bbox_imu'], np.float32)
        pc, mask = self._state_after_earlier_stages(input_record)
        lim = [float(f(gt[:, :, a])) for a in range(3) for f in (np.max, np.min)]
        inds = processing_cxx.assign3D_v2(pc.reshape(-1, 3), gt.reshape(-1, 24), gt.mean(axis=1).reshape(-1, 3),
                                          np.full((len(gt), 1), 100, np.float32), mask.reshape(-1, 1),
                                          np.zeros((pc.shape[0] * pc.shape[1], 1), np.float32), *lim, 20.0)
        input_record['bbox3d_ind_of_each_pt'] = inds.reshape((self.height, self.width, 1)).copy()


class GenerateTarget(DetectionAugmentation):
    """rangedet/core/input.py:323-519 builds the TRAINING regression / classification targets; training is outside the
    inference path this package implements, so the stage can be constructed but not applied."""

    def __init__(self, param):
        self.param = param

    def apply(self, input_record):
        raise NotImplementedError("GenerateTarget produces training targets; rangedet_amd implements the inference path")


class PadData(DetectionAugmentation):
    """rangedet/core/input.py:522-544: zero pad on the bottom / right to (pad_short, pad_long)."""

    def __init__(self, param):
        self.pad_name_list, self.pad_short, self.pad_long = param.pad_name_list, param.pad_short, param.pad_long

    def apply(self, input_record):
        _stage(input_record, "PadData", names=list(self.pad_name_list), pad_hw=(self.pad_short, self.pad_long))


class TransposeData(DetectionAugmentation):
    """rangedet/core/input.py:547-558."""

    def __init__(self, param):
        self.transpose_name_dict = param.transpose_name_dict

    def apply(self, input_record):
        _stage(input_record, "TransposeData", axes={k: tuple(v) for k, v in self.transpose_name_dict.items()})


class GenerateFPNTarget(DetectionAugmentation):
    """rangedet/core/input.py:561-607: per stride the range-interval mask and the column sampling s//2::s."""

    def __init__(self, param):
        self.interval, self.fpn_strides = param.interval, param.fpn_strides
        self.name_list, self.name_list_without_mask = param.name_list, param.name_list_without_mask

    def apply(self, input_record):
        _stage(input_record, "GenerateFPNTarget", interval=dict(self.interval), strides=tuple(self.fpn_strides),
               masked=list(self.name_list or []), plain=list(self.name_list_without_mask or []))


class TransAndReshape(DetectionAugmentation):
    """rangedet/core/input.py:610-624 -- and the point where the recorded chain runs (one rd_input_transform launch)."""

    def __init__(self, param):
        self.name_list = param.name_list

    def apply(self, input_record):
        _stage(input_record, "TransAndReshape", names=list(self.name_list))
        out = execute_chain([input_record])
        for k, v in out.items():
            input_record[k] = v[0]                      # per-record views of the (1, ...) device tensors


# ---- the fused execution of a recorded chain -----------------------------------------------------------------------------------
TEST_CHAIN = ["LoadRecord", "ProcessMissValue", "SepAndClipData", "GetUnnormalizedRange", "NormData", "GetCoordinates",
              "CombineData", "PadData", "TransposeData", "GenerateFPNTarget", "TransAndReshape"]
_COMBINE = ['range_value', 'intensity', 'elongation', 'pc_vehicle_frame_x', 'pc_vehicle_frame_y', 'pc_vehicle_frame_z',
            'inclination', 'azimuth']


def _describe(rec):
    """Validate a record's chain against what the fused kernel computes and reduce it to (clip, norm, interval, pad_hw)."""
    chain = rec.get(_CHAIN)
    if not chain or [n for n, _ in chain] != TEST_CHAIN:
        raise NotImplementedError("the device transform runs the reference's test-time chain %s; recorded: %s" %
                                  (TEST_CHAIN, [n for n, _ in (chain or [])]))
    p = dict(chain)
    if tuple(p["ProcessMissValue"]["range_fill"]) != (80, 0, 0, -1) or tuple(p["ProcessMissValue"]["pc_fill"]) != (0, 0, 0):
        raise NotImplementedError("ProcessMissValue fill values other than [80,0,0,-1] / [0,0,0]")
    if p["CombineData"]["combine"] != {'input_data': _COMBINE}:
        raise NotImplementedError("CombineData: the kernel writes the 8 channels %s as input_data" % _COMBINE)
    fp = p["GenerateFPNTarget"]
    if fp["strides"] != (1, 2, 4) or fp["masked"] != ['range_image_mask'] or sorted(fp["plain"]) != ['coord', 'pc_vehicle_frame']:
        raise NotImplementedError("GenerateFPNTarget: strides (1,2,4), mask on range_image_mask, plain pc_vehicle_frame / coord")
    need = {'input_data', 'range_image_mask', 'pc_vehicle_frame', 'unnormalized_range', 'coord'}
    if set(p["PadData"]["names"]) != need or any(tuple(a) != (2, 0, 1) for a in p["TransposeData"]["axes"].values()) or \
            set(p["TransposeData"]["axes"]) != need:
        raise NotImplementedError("PadData / TransposeData must cover %s with axes (2,0,1)" % sorted(need))
    want = ['pc_vehicle_frame_s%d' % s for s in (1, 2, 4)] + ['range_image_mask_s%d' % s for s in (1, 2, 4)]
    if sorted(p["TransAndReshape"]["names"]) != sorted(want):
        raise NotImplementedError("TransAndReshape names %s" % p["TransAndReshape"]["names"])
    return p["SepAndClipData"]["clip"], p["NormData"]["norm"], fp["interval"], tuple(p["PadData"]["pad_hw"])


_TRANSFORMS = {}


def execute_chain(records, lib=None, alloc=None):
    """ONE rd_input_transform launch for a list of records whose chains were recorded by the stage classes above.
    Returns the dict of named (B, ...) float32 device tensors (rangedet_amd.input_transform.DeviceInputTransform)."""
    from ..input_transform import DeviceInputTransform
    desc = [_describe(r) for r in records]
    if any(d != desc[0] for d in desc[1:]):
        raise ValueError("records of one batch must share one chain description")
    clip, norm, interval, pad_hw = desc[0]
    key = (repr(sorted(clip.items())), repr(sorted(norm.items())), repr(sorted(interval.items())), pad_hw, id(lib), id(alloc))
    if key not in _TRANSFORMS:
        _TRANSFORMS[key] = DeviceInputTransform(pad_hw=pad_hw, lib=lib, alloc=alloc, clip=clip, norm=norm, interval=interval)
    return _TRANSFORMS[key](records)


def run_chain(transform, records, lib=None, alloc=None):
    """Apply a transform list (the config's `transform`) to a batch of records: host-side stages per record, the image
    stages as one device launch for the whole batch.  Returns (records, named device tensors with batch dim)."""
    last = transform[-1] if transform else None
    if not isinstance(last, TransAndReshape):
        raise NotImplementedError("run_chain expects the test-time chain ending in TransAndReshape")
    for rec in records:
        for t in transform[:-1]:
            t.apply(rec)
        _stage(rec, "TransAndReshape", names=list(last.name_list))
    return records, execute_chain(records, lib, alloc)
