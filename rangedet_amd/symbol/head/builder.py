"""``RangeRCNN`` / ``RangeRpnHead`` -- the inference half of rangedet/symbol/head/builder.py with the reference's
method surface: RangeRCNN.get_test_symbol :54-77; RangeRpnHead.sep_level_type :99-154, get_fpn_output :198-266,
get_fpn_prediction :424-479, get_prediction_of_one_type :481-534.  The training half (get_train_symbol,
get_fpn_loss, get_iou_target, loss symbols) is out of scope and raises NotImplementedError.
"""
from ... import mx
from ... import mxnext as X
from ...mxnext.complicate import normalizer_factory


class RangeRCNN(object):
    def __init__(self, pDet):
        self.p = pDet
        self.fpn_strides = pDet.fpn_strides
        self.class_names = pDet.class_names

    def get_train_symbol(self, backbone, rpn_head):
        raise NotImplementedError("training graph is outside the hot path this package implements")

    def get_test_symbol(self, backbone, rpn_head):
        data = X.var('input_data')
        pcs = [X.var('pc_vehicle_frame_s{}'.format(s)) for s in self.fpn_strides]
        masks = [X.var('range_image_mask_s{}'.format(s)) for s in self.fpn_strides]
        rec_id, gt_bbox, gt_class = X.var('rec_id'), X.var('gt_bbox_imu'), X.var('gt_class')  # passed through
        feats = backbone.get_rpn_feature(data)
        assert isinstance(feats, list)
        outs = rpn_head.get_fpn_prediction(feats, pcs, masks)
        return X.group([rec_id, *outs, gt_bbox, gt_class])


class RangeRpnHead(object):
    def __init__(self, pRpn):
        self.p = pRpn
        self._prefix = ""
        self.fp16 = pRpn.fp16
        self._cls_logit = None
        self._bbox_delta = None
        self.batch_size = pRpn.batch_image
        self.class_names = pRpn.class_names
        self.fpn_strides = pRpn.fpn_strides
        self.num_classes = pRpn.num_classes
        self.num_reg_delta = pRpn.num_reg_delta
        loss = getattr(pRpn, "loss", None)  # training-only weights; kept when present for attribute parity
        self.cls_loss_weight = getattr(loss, "cls_loss_weight", None)
        self.reg_loss_weight = getattr(loss, "reg_loss_weight", None)
        self.scale_loss_shift = getattr(pRpn, "scale_loss_shift", 1.0) if self.fp16 else 1.0

    # ---- heads ---------------------------------------------------------------------------------------------
    def _tower(self, norm, feat, kind, level, layers, channel):
        for i in range(layers):
            feat = X.convnormrelu(norm, feat, kernel=3, filter=channel,
                                  name=self._prefix + 'rpn_{}_conv_{}_lvl_{}'.format(kind, i, level), no_bias=True,
                                  init=X.gauss(0.01))
        return feat

    def get_fpn_output(self, conv_feat_list):
        h = self.p.head
        norm = self.p.normalizer if hasattr(self.p, 'normalizer') else normalizer_factory(type='local', ndev=None, mom=0.9)
        self._cls_logit, self._bbox_delta = [], []
        for level, feat in enumerate(conv_feat_list):
            cls_feat = self._tower(norm, feat, "cls", level, h.cls_conv_layers, h.cls_conv_channel)
            reg_feat = self._tower(norm, feat, "reg", level, h.reg_conv_layers, h.reg_conv_channel)
            logit = X.conv(cls_feat, filter=self.num_classes, name=self._prefix + 'rpn_cls_logit_lvl_' + str(level),
                           no_bias=False, init=X.gauss(0.01))
            delta = X.conv(reg_feat, filter=self.num_reg_delta * self.num_classes,
                           name=self._prefix + 'rpn_reg_delta_lvl_' + str(level), no_bias=False, init=X.gauss(0.01))
            if self.fp16:
                logit = X.to_fp32(logit, 'rpn_cls_logit_lvl_{}_fp32'.format(level))
                delta = X.to_fp32(delta, 'rpn_reg_delta_lvl_{}_fp32'.format(level))
            self._cls_logit.append(logit)
            self._bbox_delta.append(delta)
        return self._cls_logit, self._bbox_delta

    def sep_level_type(self, cls_logit_list, bbox_delta_list, concat_all_level_per_class=False):
        cls_d = {c: [] for c in self.class_names}
        reg_d = {c: [] for c in self.class_names}
        for level, (logit, delta) in enumerate(zip(cls_logit_list, bbox_delta_list)):
            score = X.reshape(logit, shape=(self.batch_size, self.num_classes, -1),
                              name='cls_score_reshape_lvl_{}'.format(level))
            delta = X.reshape(delta, shape=(self.batch_size, self.num_classes, self.num_reg_delta, -1),
                              name='bbox_delta_reshape_lvl_{}'.format(level))
            for i, cname in enumerate(self.class_names[:self.num_classes]):
                s = mx.squeeze(mx.slice_axis(score, axis=1, begin=i, end=i + 1), axis=1)
                cls_d[cname].append(s)
                d = mx.squeeze(mx.slice_axis(delta, axis=1, begin=i, end=i + 1), axis=1)
                reg_d[cname].append(X.transpose(d, (0, 2, 1)))
        if concat_all_level_per_class:
            cls_d = {c: mx.concat(*v, dim=1) for c, v in cls_d.items()}
            reg_d = {c: mx.concat(*v, dim=1) for c, v in reg_d.items()}
        return cls_d, reg_d

    def get_fpn_loss(self, *args, **kwargs):
        raise NotImplementedError("training graph is outside the hot path this package implements")

    get_iou_target = get_vfl_loss = get_normalize_reg_loss = get_fpn_loss

    # ---- prediction ------------------------------------------------------------------------------------------
    def get_fpn_prediction(self, conv_feat_list, pc_vehicle_frame_list, range_image_mask_list):
        logits, deltas = self.get_fpn_output(conv_feat_list)
        logit_d, delta_d = self.sep_level_type(logits, deltas, concat_all_level_per_class=True)
        score_d = {k: X.sigmoid(v, name=k + '_sigmoid') for k, v in logit_d.items()}
        all_pc = mx.concat(*pc_vehicle_frame_list, dim=1)
        all_mask = mx.concat(*range_image_mask_list, dim=1)
        prop = self.p.all_proposal
        outs = []
        for c in self.class_names:
            outs += list(self.get_prediction_of_one_type(score_d[c], delta_d[c], all_pc, all_mask, prop.nms_thr[c],
                                                         prop.rpn_pre_nms_top_n[c], prop.rpn_post_nms_top_n[c]))
        return outs

    def get_prediction_of_one_type(self, cls_score, bbox_delta, pc_vehicle_frame, mask, nms_thr, pre_nms_top_n,
                                   post_nms_top_n):
        fg_score, fg_delta, fg_pc = mx.Custom(cls_score=cls_score, bbox_delta=bbox_delta, pc=pc_vehicle_frame,
                                              mask=mask, op_type="get_sorted_foreground", num_fgs=pre_nms_top_n,
                                              name=self._prefix + "get_foreground")
        boxes = mx.contrib.Decode3DBbox(fg_delta, fg_pc, is_bin=False)
        if getattr(self.p, 'wnms', False):
            return fg_score, boxes, mx.zeros(shape=(1,))
        keep_inds, final = mx.contrib.NMS3D(boxes, nms_thr, post_nms_top_n)
        return fg_score, final, keep_inds
