"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PyTorch-CPU float32 restatement of the RangeDet *test* graph, op for op as the reference builds it with MXNet
symbols (citations relative to /root/reference):
  rangedet/symbol/backbone/meta_kernel.py:16-38,76-103,105-164,166-240   Meta-Kernel (un-fused: im2col, reshape,
                                                                          broadcast_minus, two 1x1 convs, multiply)
  rangedet/symbol/backbone/dla_backbone.py:18-56,59-103,106-161          DLA backbone
  rangedet/symbol/head/builder.py:99-154,198-266,424-534                 heads, per-class flatten, sigmoid, top-k, decode
  mxnext/simple.py:123-158,545-580 ; mxnext/complicate.py:14,26-45       conv / deconv wrappers, BN eps = 1e-5 + 1e-10

PARITY UNPINNED: the arithmetic of mx.sym.Convolution / Deconvolution / BatchNorm / im2col / topk is third-party
(mxnet==2.0.0 is not installed and not vendored) and the reference has no tests or golden vectors for it.  This file
relies on the documented definitions: im2col channel order c*kh*kw + ki*kw + kj (== torch unfold), Convolution =
cross-correlation NCHW/OIHW with zero padding, Deconvolution weight (I,O,kh,kw) with out = (in-1)*s - 2p + k
(== conv_transpose2d), BatchNorm inference gamma*(x-mean)/sqrt(var+eps)+beta.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import cpu_ops

EPS = 1e-5 + 1e-10  # mxnext/complicate.py:14


class Cfg:
    """The values of config/rangedet/rangedet_veh_wo_aug_4_18e.py that shape the test graph."""
    fpn_strides = (1, 2, 4)
    num_block = {'res1': 2, 'res2a': 3, 'res2': 3, 'res3a': 5, 'res3': 5, 'agg1': 2, 'agg2': 2, 'agg2a': 1, 'agg3': 2}
    num_filter = {'res1': 64, 'res2a': 64, 'res2': 128, 'res3a': 128, 'res3': 128, 'agg1': 64, 'agg2': 128,
                  'agg2a': 64, 'agg3': 64}
    meta_kernel_units = {'res1_unit2': dict(stride=1, data_channels=64, coord_channels=3, channel_list=[32, 64], kernel_size=3)}
    add_data_sc = True
    head_layers = 4
    head_channel = 128
    num_classes = 1
    num_reg_delta = 8
    class_names = ('veh',)
    pre_nms_top_n = {'veh': 50000}
    min_score = {'veh': 0.5, 'ped': 0.4, 'cyc': 0.3}     # config:332 (TestParam.min_score)
    thr_lo, thr_hi, is_3d_iou = 0.1, 0.5, False


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def bn(x, P, name):
    g, b, m, v = (T(P[name + s]) for s in ("_gamma", "_beta", "_moving_mean", "_moving_var"))
    sh = (1, -1, 1, 1)
    return g.view(sh) * (x - m.view(sh)) / torch.sqrt(v.view(sh) + EPS) + b.view(sh)


def conv(x, P, name, kernel=1, stride=(1, 1), bias=False):
    pad = ((kernel - 1) + 1) // 2 if kernel > 1 else 0  # mxnext/simple.py:131-135 (dilate 1)
    b = T(P[name + "_bias"]) if bias else None
    return F.conv2d(x, T(P[name + "_weight"]), b, stride=stride, padding=pad)


def meta_kernel_unit(data, coord, P, name):
    """meta_baseline_bias (meta_kernel.py:166-240) then BN/ReLU/1x1/BN/ReLU (dla_backbone.py:92-97)."""
    B, C, H, W = data.shape
    pre = name + "_"
    Wn = str(W)
    cs = F.unfold(coord, 3, padding=1).view(B, 3, 9, H, W)           # sample_coord + reshape
    rel = cs - coord.unsqueeze(2)                                      # broadcast_minus
    x = rel.reshape(B, 3, 9 * H, W)                                    # mlp(): reshape (B, C, -1, W)
    x = F.conv2d(x, T(P[pre + Wn + "_mlp0_weight"]), T(P[pre + Wn + "_mlp0_bias"]))
    x = F.relu(x)
    x = F.conv2d(x, T(P[pre + Wn + "_mlp1_weight"]), T(P[pre + Wn + "_mlp1_bias"]))
    wts = x.view(B, 64, 9, H, W)
    ds = F.unfold(data, 3, padding=1).view(B, C, 9, H, W)
    out = (ds * wts).reshape(B, C * 9, H, W)
    y = F.relu(bn(out, P, name + "point_wise_mlp_bn1"))
    y = conv(y, P, name + "aggregation_conv1", 1)
    return F.relu(bn(y, P, name + "aggregation_bn1"))


def basicblock(x, coord, P, cfg, name, stride, proj):
    if name in cfg.meta_kernel_units:
        r1 = meta_kernel_unit(x, coord, P, name)
    else:
        r1 = F.relu(bn(conv(x, P, name + "_conv1", 3), P, name + "_bn1"))
    b2 = bn(conv(r1, P, name + "_conv2", 3, stride), P, name + "_bn2")
    sc = bn(conv(x, P, name + "_sc", 1, stride), P, name + "_sc_bn") if proj else x
    return F.relu(b2 + sc)


def res_stage(x, coord, P, cfg, name, nblk, stride):
    x = basicblock(x, coord, P, cfg, name + "_unit1", stride, True)
    for i in range(2, nblk + 1):
        x = basicblock(x, coord, P, cfg, "%s_unit%d" % (name, i), (1, 1), False)
    return x


def agg_stage(const, up, coord, P, cfg, name, nblk, k, s, p):
    u = F.conv_transpose2d(up, T(P[name + "_deconv_weight"]), None, stride=s, padding=p)
    u = F.relu(bn(u, P, name + "_deconv_bn"))
    return res_stage(const + u, coord, P, cfg, name + "_res", nblk, (1, 1))


def backbone(data, coord, P, cfg=Cfg):
    nb = cfg.num_block
    res1 = res_stage(data, coord, P, cfg, 'res1', nb['res1'], (1, 1))
    res2a = res_stage(res1, coord, P, cfg, 'res2a', nb['res2a'], (1, 2))
    res2 = res_stage(res2a, coord, P, cfg, 'res2', nb['res2'], (1, 2))
    res3a = res_stage(res2, coord, P, cfg, 'res3a', nb['res3a'], (1, 2))
    res3 = res_stage(res3a, coord, P, cfg, 'res3', nb['res3'], (1, 2))
    agg2 = agg_stage(res2, res3, coord, P, cfg, "agg2", nb['agg2'], (3, 8), (1, 4), (1, 2))
    agg1 = agg_stage(res1, res2, coord, P, cfg, "agg1", nb['agg1'], (3, 8), (1, 4), (1, 2))
    agg2a = agg_stage(res2a, agg2, coord, P, cfg, "agg2a", nb['agg2a'], (3, 4), (1, 2), (1, 1))
    agg3 = agg_stage(agg1, agg2a, coord, P, cfg, "agg3", nb['agg3'], (3, 4), (1, 2), (1, 1))
    if cfg.add_data_sc:
        agg3 = torch.cat([data, agg3], 1)
    d = {1: agg3, 2: agg2a, 4: agg2, 16: res3}
    return [d[s] for s in cfg.fpn_strides], dict(res1=res1, res2a=res2a, res2=res2, res3a=res3a, res3=res3,
                                                 agg2=agg2, agg1=agg1, agg2a=agg2a)


def head(feats, P, cfg=Cfg, per_class=False):
    """get_fpn_output + sep_level_type(concat) (builder.py:198-266,99-154): logits (B,N), deltas (B,N,8) of class 0,
    or with per_class a list of them, one per class."""
    logits, deltas = [], []
    for lvl, f in enumerate(feats):
        c = r = f
        for i in range(cfg.head_layers):
            n = 'rpn_cls_conv_%d_lvl_%d' % (i, lvl)
            c = F.relu(bn(conv(c, P, n, 3), P, n + "_bn"))
        for i in range(cfg.head_layers):
            n = 'rpn_reg_conv_%d_lvl_%d' % (i, lvl)
            r = F.relu(bn(conv(r, P, n, 3), P, n + "_bn"))
        lg = conv(c, P, 'rpn_cls_logit_lvl_%d' % lvl, 1, bias=True)
        dl = conv(r, P, 'rpn_reg_delta_lvl_%d' % lvl, 1, bias=True)
        B = lg.shape[0]
        logits.append(lg.reshape(B, cfg.num_classes, -1))
        deltas.append(dl.reshape(B, cfg.num_classes, cfg.num_reg_delta, -1).transpose(2, 3))
    lg, dl = torch.cat(logits, 2), torch.cat(deltas, 2)       # (B, classes, N), (B, classes, N, 8)
    if per_class:
        return [(lg[:, i], dl[:, i]) for i in range(cfg.num_classes)]   # builder.py:134-142: one slice per class
    return lg[:, 0], dl[:, 0]


def forward(inputs, P, cfg=Cfg, num_fgs=None, stages=False):
    """The test symbol (builder.py:54-77): returns fg_cls_score (B,k), decoded_bbox (B,k,10) [+ intermediates]."""
    with torch.no_grad():
        data = T(inputs["input_data"])
        coord = T(inputs["coord_s1"])
        feats, inter = backbone(data, coord, P, cfg)
        logit, delta = head(feats, P, cfg)
        score = torch.sigmoid(logit)
        pc = np.concatenate([inputs["pc_vehicle_frame_s%d" % s] for s in cfg.fpn_strides], 1)
        mask = np.concatenate([inputs["range_image_mask_s%d" % s] for s in cfg.fpn_strides], 1)
        k = num_fgs or cfg.pre_nms_top_n[cfg.class_names[0]]
        if isinstance(k, dict):
            k = k[cfg.class_names[0]]
        sc, dl, pp, idx = cpu_ops.get_sorted_foreground(score.numpy(), delta.numpy(), pc, mask, k)
        boxes = cpu_ops.decode3d(dl, pp, False)
    out = dict(fg_cls_score=sc, decoded_bbox=boxes, logit=logit.numpy(), delta=delta.numpy(), sorted_idx=idx)
    if cfg.num_classes > 1:   # builder.py:467-478: the same three ops per class, each with its own top-k
        out["classes"] = {}
        with torch.no_grad():
            for cname, (lg_c, dl_c) in zip(cfg.class_names, head(feats, P, cfg, per_class=True)):
                kc = (num_fgs or cfg.pre_nms_top_n)[cname] if isinstance(num_fgs or cfg.pre_nms_top_n, dict) else num_fgs
                s_c, d_c, p_c, i_c = cpu_ops.get_sorted_foreground(torch.sigmoid(lg_c).numpy(), dl_c.numpy(), pc, mask, kc)
                out["classes"][cname] = dict(fg_cls_score=s_c, decoded_bbox=cpu_ops.decode3d(d_c, p_c, False),
                                             logit=lg_c.numpy(), delta=dl_c.numpy(), sorted_idx=i_c)
    if stages:
        out["feats"] = [f.numpy() for f in feats]
        out["inter"] = {k_: v.numpy() for k_, v in inter.items()}
    return out


def postprocess(fg_cls_score, decoded_bbox, cfg=Cfg, order=None, cls=None):
    """tools/test.py:184-224 for one frame: score filter, 10->11 dim, wnms_4c, 12->8 dim (cls: the class whose min_score applies)."""
    cn = cls or cfg.class_names[0]
    dets = cpu_ops.score_filter_to_dets(fg_cls_score, decoded_bbox, cfg.min_score[cn])
    if dets.shape[0] == 0:
        return dets, np.zeros((0, 12), np.float32), [], np.zeros((0, 8))
    flat, keep = cpu_ops.wnms_4c(dets, cfg.thr_lo, cfg.thr_hi, cfg.is_3d_iou, 100, order=order)
    rows = np.array(flat, dtype=np.float32).reshape(-1, 12)
    return dets, rows, keep, cpu_ops.bbox3d_12dim_to_8dim(np.array(flat).reshape(-1, 12))
