"""Inference-side values of the reference config ``config/rangedet/rangedet_veh_wo_aug_4_18e.py`` with the same
param-class and attribute names (General, KvstoreParam, BackboneParam, RpnParam, DetParam, TestParam, ModelParam ...
cited lines: :30-161 model params, :198-215 TestParam).  ``get_config(is_train=False)`` builds the test symbol at call
time exactly like the reference does at import (:153-161).  Training-only classes (OptimizeParam, AugParam, target
generation) are out of scope; ``get_config(True)`` raises.

The ped / all_36e variants differ only in class + sampling + epochs; ``variant`` covers them.
"""
from ..core import detection_metric as metric
from ..core.input import (CombineData, FilterGTClass, GenerateFPNTarget, GetCoordinates, GetUnnormalizedRange, LoadGTInfo,
                          LoadRecord, NormData, PadData, ProcessMissValue, SepAndClipData, TransAndReshape, TransposeData)
from ..mxnext.complicate import normalizer_factory
from ..symbol.backbone.dla_backbone import DLABackbone as Backbone
from ..symbol.head.builder import RangeRCNN as Detector
from ..symbol.head.builder import RangeRpnHead as RpnHead

_VARIANTS = {
    "veh": dict(label_set=[1], class_names=('veh',), filter_class=['TYPE_VEHICLE']),
    "ped": dict(label_set=[2], class_names=('ped',), filter_class=['TYPE_PEDESTRIAN']),
    # BASELINE config 5: KITTI range images (datasets/create_range_image_in_kitti.py:121,126: 64 x 2048 x 5 = range, x, y, z,
    # intensity) with a mixed vehicle + pedestrian head (builder.py:134-142,467-478 per-class slicing / top-k).  The
    # reference ships no KITTI config file; call get_config(variant="kitti", feat_size=(64, 2048), pad_field=(64, 2048)).
    "kitti": dict(label_set=[1, 2], class_names=('veh', 'ped'), filter_class=['Car', 'Pedestrian']),
}
KITTI_INPUT_CHANNELS = 5


def variant_classes(variant):
    return _VARIANTS[variant]["class_names"]


def get_config(is_train=False, variant="veh", feat_size=(64, 2650), pad_field=(64, 2656), fp16=True, batch_image=1,
               pre_nms_top_n=None, wnms=True, sampling_rate=4, end_epoch=18, name=None, backbone=None):
    """backbone: optional {attribute: value} overrides of BackboneParam (dict-valued attributes are merged key by key), e.g.
    backbone={'num_filter': {'res1': 96}} -- the reference's config surface lets these vary (dla_backbone.py:59-103,130-161); the
    symbol builds for any value, the HIP lowering only takes the shipped widths and says so (lower.py)."""
    _wnms = bool(wnms)
    if is_train:
        raise NotImplementedError("training is outside the hot path this package implements")
    V = _VARIANTS[variant]
    _bi, _fs, _pf, _fp16, _sr, _ee, _name = batch_image, feat_size, pad_field, fp16, sampling_rate, end_epoch, name

    class General:
        batch_image = _bi
        log_frequency = 100
        name = _name or __name__.rsplit(".")[-1]
        fp16 = _fp16
        scale_loss_shift = 128
        feat_size = _fs
        label_set = V["label_set"]
        num_classes = len(label_set)
        class_names = V["class_names"]
        pad_field = _pf

    class KvstoreParam:
        sync_flag = True
        kvstore = "local"
        use_horovod = True
        gpus = [0, 1, 2, 3, 4, 5, 6, 7]
        batch_image = General.batch_image
        fp16 = General.fp16

    class NormalizeParam:
        normalizer = normalizer_factory(type="localbn", ndev=len(KvstoreParam.gpus))

    class FpnParam:
        fpn_strides = (1, 2, 4)
        strategy = 'range'
        interval = {1: (30, 100), 2: (15, 30), 4: (0, 15)}
        name_list = ['range_image_mask']
        name_list_without_mask = ['pc_vehicle_frame', 'coord']

    class BackboneParam:
        fp16 = General.fp16
        normalizer = NormalizeParam.normalizer
        fpn_strides = FpnParam.fpn_strides
        batch_image = General.batch_image
        range_image_shape_hw = General.pad_field
        meta_kernel_units = {'res1_unit2': dict(stride=1, meta_func_param='meta_baseline_bias', data_channels=64,
                                                coord_channels=3, channel_list=[32, 64], kernel_size=3)}
        num_block = {'res1': 2, 'res2a': 3, 'res2': 3, 'res3a': 5, 'res3': 5, 'agg1': 2, 'agg2': 2, 'agg2a': 1, 'agg3': 2}
        num_filter = {'res1': 64, 'res2a': 64, 'res2': 128, 'res3a': 128, 'res3': 128, 'agg1': 64, 'agg2': 128,
                      'agg2a': 64, 'agg3': 64}
        add_data_sc = True

    for _k, _v in (backbone or {}).items():
        _cur = getattr(BackboneParam, _k)
        setattr(BackboneParam, _k, dict(_cur, **_v) if isinstance(_cur, dict) and isinstance(_v, dict) else _v)

    class RpnParam:
        fp16 = General.fp16
        normalizer = NormalizeParam.normalizer
        batch_image = General.batch_image
        feat_size = General.feat_size
        scale_loss_shift = General.scale_loss_shift
        class_names = General.class_names
        num_classes = General.num_classes
        fpn_strides = FpnParam.fpn_strides
        num_reg_delta = 8
        wnms = _wnms      # config:155 sets True; False = the contrib.NMS3D branch (builder.py:530-534)

        class head:
            cls_conv_layers = 4
            cls_conv_channel = 128
            reg_conv_layers = 4
            reg_conv_channel = 128

        class all_proposal:
            rpn_pre_nms_top_n = dict({'veh': 50000, 'ped': 5000, 'cyc': 5000}, **(pre_nms_top_n or {}))
            rpn_post_nms_top_n = {'veh': 200, 'ped': 200, 'cyc': 100}
            nms_thr = {'veh': 0.2, 'ped': 0.2, 'cyc': 0.2}

    class RoiParam:
        pass

    class BboxParam:
        pass

    class DetParam:
        fpn_strides = FpnParam.fpn_strides
        class_names = General.class_names

    class DatasetParam:
        image_set = 'validation'
        sampling_rate = _sr
        filter_class = V["filter_class"]

    backbone = Backbone(BackboneParam)
    rpn_head = RpnHead(RpnParam)
    detector = Detector(DetParam)
    test_sym = detector.get_test_symbol(backbone, rpn_head)

    class ModelParam:
        train_symbol = None
        test_symbol = test_sym
        from_scratch = True

    class OptimizeParam:
        class schedule:
            end_epoch = _ee

    class TestParam:
        min_score = {'veh': 0.5, 'ped': 0.4, 'cyc': 0.3}
        max_det_per_image = 100
        class_names = General.class_names

        class model:
            prefix = "experiments/{}/checkpoint".format(General.name)
            epoch = OptimizeParam.schedule.end_epoch

        class nms:
            wnms = bool(getattr(RpnParam, 'wnms', False))
            thr_lo = 0.1
            thr_hi = 0.5
            is_3d_iou = False

    class LabelMapParam:
        mapping = {1: 1, 2: 2, 3: 3, 4: 4, 0: 5}
        test_mapping = {0: 1, 1: 2, 2: 4}

    class ClipDataParam:                                  # config:245-255
        clip_data_dict = {'range_value': (0, 80), 'intensity': (0, 1), 'elongation': (0, 1), 'pc_vehicle_frame_x': (-80, 80),
                          'pc_vehicle_frame_y': (-80, 80), 'pc_vehicle_frame_z': (-5, 10), 'inclination': (-0.5, 0.1),
                          'azimuth': (-6.283185307179586, 1.5707963267948966)}

    class NormDataParam:                                  # config:257-267 (mean, variance)
        norm_data_dict = {'range_value': (20.0, 1500.0), 'intensity': (0.1, 0.01), 'elongation': (7.2558375e-02, 2.6764875e-02),
                          'pc_vehicle_frame_x': (1.5672500e+00, 3.0740625e+02), 'pc_vehicle_frame_y': (9.8824875e-01, 2.1913250e+02),
                          'pc_vehicle_frame_z': (1.4, 1.0), 'inclination': (-8.8427375e-02, 9.9001750e-03),
                          'azimuth': (-7.8061250e-03, 2.5494125e+00)}

    class CombineDataParam:                               # config:269-282
        combine_name_dict = {'input_data': ['range_value', 'intensity', 'elongation', 'pc_vehicle_frame_x', 'pc_vehicle_frame_y',
                                            'pc_vehicle_frame_z', 'inclination', 'azimuth']}

    class PadDataParam:                                   # config:291-314 (test branch)
        pad_short, pad_long = General.pad_field
        pad_name_list = ['input_data', 'range_image_mask', 'pc_vehicle_frame', 'unnormalized_range', 'coord']

    class TransposeDataParam:                             # config:316-334 (test branch)
        transpose_name_dict = {n: (2, 0, 1) for n in PadDataParam.pad_name_list}

    class TransAndReshapeParam:                           # config:336-341 (test branch)
        name_list = ['pc_vehicle_frame_s1', 'pc_vehicle_frame_s2', 'pc_vehicle_frame_s4',
                     'range_image_mask_s1', 'range_image_mask_s2', 'range_image_mask_s4']

    # the reference's test-time chain (config:380-399), built from this package's classes: the image-sized stages run as one
    # fused device launch (rangedet_amd.core.input)
    transform = [LoadRecord(), LoadGTInfo(), FilterGTClass(General.label_set), ProcessMissValue(), SepAndClipData(ClipDataParam),
                 GetUnnormalizedRange(), NormData(NormDataParam), GetCoordinates(), CombineData(CombineDataParam),
                 PadData(PadDataParam), TransposeData(TransposeDataParam), GenerateFPNTarget(FpnParam),
                 TransAndReshape(TransAndReshapeParam)]
    pc_stride = ["pc_vehicle_frame_s{}".format(s) for s in RpnParam.fpn_strides]
    mask_stride = ["range_image_mask_s{}".format(s) for s in RpnParam.fpn_strides]
    data_name = ["input_data", "gt_bbox_imu", "gt_class", "rec_id"] + pc_stride + mask_stride + ['coord_s1']
    label_name = []
    metric_list = [metric.ScalarLoss("L1-s{}".format(s), ["rpn_reg_loss_s{}_output".format(s)], []) for s in FpnParam.fpn_strides] + \
                  [metric.ScalarLoss("cls-s{}".format(s), ["rpn_cls_loss_s{}_output".format(s)], []) for s in FpnParam.fpn_strides]
    return General, KvstoreParam, RpnParam, RoiParam, BboxParam, DatasetParam, ModelParam, OptimizeParam, TestParam, \
        transform, data_name, label_name, metric_list, LabelMapParam
