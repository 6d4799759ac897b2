"""rangedet_amd.core.input: the reference config's transform list (config/rangedet/rangedet_veh_wo_aug_4_18e.py:380-399) built
from this package's classes under the reference's import path, run as one device launch, against the numpy restatement of the
chain (oracle/input_ref.py); plus the host-side stages and the config / alias plumbing."""
import numpy as np
import pytest

from conftest import BOTH
from oracle import input_ref as IR
from rangedet_amd import compat, synth
from rangedet_amd.config import rangedet_veh_wo_aug_4_18e as cfgmod


def test_reference_import_paths_and_config_transform_list():
    compat.install_aliases()
    import rangedet.core.detection_metric as metric
    from rangedet.core.input import (Bbox3dAssigner, CombineData, FilterGTClass, GenerateFPNTarget, GenerateTarget,  # noqa: F401
                                     GetCoordinates, GetFixedLengthGTBbox, GetUnnormalizedRange, LoadGTInfo, LoadRecord,
                                     NormData, PadData, ProcessMissValue, SepAndClipData, TransAndReshape, TransposeData)
    cfg = cfgmod.get_config(False)
    transform, data_name, label_name, metric_list = cfg[9], cfg[10], cfg[11], cfg[12]
    assert [type(t).__name__ for t in transform] == [
        "LoadRecord", "LoadGTInfo", "FilterGTClass", "ProcessMissValue", "SepAndClipData", "GetUnnormalizedRange", "NormData",
        "GetCoordinates", "CombineData", "PadData", "TransposeData", "GenerateFPNTarget", "TransAndReshape"]   # config:380-399
    assert isinstance(transform[0], LoadRecord) and transform[2].valid_class == [1]
    assert 'azimuth' not in transform[4].clip_data_dict and len(transform[4].clip_data_dict) == 7          # input.py:149
    assert transform[9].pad_short == 64 and transform[9].pad_long == 2656
    assert data_name[:4] == ["input_data", "gt_bbox_imu", "gt_class", "rec_id"] and label_name == []
    assert [m.name for m in metric_list] == ["L1-s1", "L1-s2", "L1-s4", "cls-s1", "cls-s2", "cls-s4"]     # config:403-416
    m = metric.ScalarLoss("x", ["o"], [])
    m.update([], [np.array([1.0, 3.0])])
    m.update([], [np.array([2.0])])
    assert m.get() == ("x", 3.0)
    with pytest.raises(NotImplementedError):
        GenerateTarget(object()).apply({})
    # the product's constants are the reference's (restated independently in the oracle)
    assert synth.CLIP == IR.CLIP and synth.NORM == IR.NORM and synth.INTERVAL == IR.INTERVAL and synth.COMBINE == IR.COMBINE


def test_host_side_stages():
    from rangedet_amd.core.input import EPS, FilterGTClass, GetFixedLengthGTBbox, LoadGTInfo
    rng = np.random.default_rng(0)
    rec = dict(gt_class=np.array([1, 2, 1, 4]), gt_bbox_imu=rng.standard_normal((4, 8, 3)), gt_bbox_csa=rng.standard_normal((4, 7)),
               gt_bbox_yaw=rng.standard_normal(4), points_in_box=np.arange(4), meta_data=np.zeros((4, 4)))
    box1 = rec['gt_bbox_imu'][2].copy()
    LoadGTInfo().apply(rec)
    assert all(rec[k].dtype == np.float32 for k in LoadGTInfo.KEYS)
    FilterGTClass([1]).apply(rec)
    assert rec['gt_class'].tolist() == [1, 1] and np.allclose(rec['gt_bbox_imu'][1], box1) and rec['points_in_box'].tolist() == [0, 2]
    p = type("P", (), dict(class_type=['TYPE_VEHICLE'], fixed_length=200))
    GetFixedLengthGTBbox(p).apply(rec)
    fx = rec['gt_bbox_veh_for_iou_pred']
    assert fx.shape == (200, 8) and np.allclose(fx[1], box1[:4, :2].reshape(-1), atol=1e-6)
    assert np.allclose(fx[2], [0, 0, 0, EPS, EPS, EPS, EPS, 0])
    FilterGTClass([3]).apply(rec)                                  # nothing left -> one zero box (input.py:81-86)
    assert rec['gt_class'].shape == (1,) and rec['gt_bbox_imu'].shape == (1, 8, 3) and not rec['gt_bbox_imu'].any()


@pytest.mark.parametrize("be", BOTH, indirect=True)
def test_transform_list_runs_as_one_device_launch(be, tmp_path):
    """The config's own transform list applied to records (one of them through the npz / pc_url path): equal to the numpy
    restatement of the chain, stage list unchanged; per record via TransAndReshape.apply and batched via run_chain."""
    from rangedet_amd.core import input as CI
    H, W, Wp = 8, 62, 64
    cfg = cfgmod.get_config(False, feat_size=(H, W), pad_field=(H, Wp))
    transform = cfg[9]
    raws = [synth.raw_record(i, H=H, W=W) for i in range(3)]
    f = tmp_path / "1.npz"
    np.savez(f, **raws[1])
    recs = [dict(raws[0]), dict(pc_url=str(f)), dict(raws[2])]
    for r in recs:
        r.update(gt_class=np.array([1.0]), gt_bbox_imu=np.zeros((1, 8, 3)), gt_bbox_csa=np.zeros((1, 7)), gt_bbox_yaw=np.zeros(1),
                 points_in_box=np.zeros(1), meta_data=np.zeros((1, 4)))
    CI._TRANSFORMS.clear()
    _, out = CI.run_chain(transform, recs, lib=be.lib, alloc=be.alloc)
    be.alloc.sync()
    ref = [IR.transform(r, (H, Wp)) for r in raws]
    for k in ref[0]:
        got = np.array(be.alloc.to_numpy(out[k]))
        want = np.concatenate([r[k] for r in ref], 0)
        assert got.shape == want.shape, k
        if k == "input_data":
            assert np.array_equal(got[:, :7], want[:, :7]) and np.abs(got[:, 7] - want[:, 7]).max() < 1e-6   # azimuth: atan2f
        else:
            assert np.array_equal(got, want), k
    # a chain the fused kernel does not implement is refused, not approximated
    bad = dict(raws[0])
    for t in transform[:3] + transform[4:-1]:                       # ProcessMissValue left out
        t.apply(bad)
    with pytest.raises(NotImplementedError):
        transform[-1].apply(bad)


def test_ped_and_all_36e_config_modules():
    """config/rangedet/rangedet_{ped,veh}_wo_aug_{4_18e,all_36e}.py: same graph, class / sampling / epochs differ."""
    import importlib
    want = {"rangedet_veh_wo_aug_4_18e": (('veh',), [1], 4, 18, ['TYPE_VEHICLE']),
            "rangedet_ped_wo_aug_4_18e": (('ped',), [2], 4, 18, ['TYPE_PEDESTRIAN']),
            "rangedet_veh_wo_aug_all_36e": (('veh',), [1], 1, 36, ['TYPE_VEHICLE']),
            "rangedet_ped_wo_aug_all_36e": (('ped',), [2], 1, 36, ['TYPE_PEDESTRIAN'])}
    for mod, (names, labels, sr, ep, fc) in want.items():
        cfg = importlib.import_module("rangedet_amd.config." + mod).get_config(False, feat_size=(8, 30), pad_field=(8, 32))
        G, D, T = cfg[0], cfg[5], cfg[8]
        assert (G.name, G.class_names, G.label_set, D.sampling_rate, T.model.epoch, D.filter_class) == (mod, names, labels, sr, ep, fc)
        assert T.model.prefix == "experiments/%s/checkpoint" % mod and len(cfg[6].test_symbol.list_arguments()) > 300
