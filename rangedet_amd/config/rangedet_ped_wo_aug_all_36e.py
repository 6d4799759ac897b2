"""Inference-side mirror of the reference config ``config/rangedet/rangedet_ped_wo_aug_all_36e.py``: the vehicle 4-18e config
(``rangedet_veh_wo_aug_4_18e``) with class 'ped', DatasetParam.sampling_rate 1 and 36 epochs -- the only lines in which the
reference files differ (label_set / class_names :10-13, sampling_rate :64, end_epoch :188)."""
from . import rangedet_veh_wo_aug_4_18e as _base


def get_config(is_train=False, **kw):
    kw.setdefault("variant", 'ped')
    return _base.get_config(is_train, sampling_rate=1, end_epoch=36, name=__name__.rsplit(".")[-1], **kw)
