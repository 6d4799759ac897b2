"""Inference-side values of the reference config ``config/rangedet/rangedet_veh_wo_aug_4_18e.py`` with the same
param-class and attribute names (General, KvstoreParam, BackboneParam, RpnParam, DetParam, TestParam, ModelParam ...
cited lines: :30-161 model params, :198-215 TestParam).  ``get_config(is_train=False)`` builds the test symbol at call
time exactly like the reference does at import (:153-161).  Training-only classes (OptimizeParam, AugParam, target
generation) are out of scope; ``get_config(True)`` raises.

The ped / all_36e variants differ only in class + sampling + epochs; ``variant`` covers them.
"""
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


def get_config(is_train=False, variant="veh", feat_size=(64, 2650), pad_field=(64, 2656), fp16=True, batch_image=1,
               pre_nms_top_n=None, wnms=True):
    _wnms = bool(wnms)
    if is_train:
        raise NotImplementedError("training is outside the hot path this package implements")
    V = _VARIANTS[variant]
    _bi, _fs, _pf, _fp16 = batch_image, feat_size, pad_field, fp16

    class General:
        batch_image = _bi
        log_frequency = 100
        name = __name__.rsplit(".")[-1]
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
        sampling_rate = 4
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
            end_epoch = 18

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

    pc_stride = ["pc_vehicle_frame_s{}".format(s) for s in RpnParam.fpn_strides]
    mask_stride = ["range_image_mask_s{}".format(s) for s in RpnParam.fpn_strides]
    data_name = ["input_data", "gt_bbox_imu", "gt_class", "rec_id"] + pc_stride + mask_stride + ['coord_s1']
    transform, label_name, metric_list = [], [], []
    return General, KvstoreParam, RpnParam, RoiParam, BboxParam, DatasetParam, ModelParam, OptimizeParam, TestParam, \
        transform, data_name, label_name, metric_list, LabelMapParam
