"""``rangedet.core.detection_metric`` for the inference package: the config builds ``metric.ScalarLoss(name, output_names,
label_names)`` objects into ``metric_list`` (config/rangedet/rangedet_veh_wo_aug_4_18e.py:403-416); they are consumed by the
training loop only (out of scope here), so this module provides the class with MXNet's EvalMetric bookkeeping
(``update`` / ``get`` / ``reset``, rangedet/core/detection_metric.py) on numpy values instead of ``mx.nd`` arrays."""
import numpy as np


class EvalMetric:
    def __init__(self, name, output_names=None, label_names=None):
        self.name, self.output_names, self.label_names = str(name), output_names, label_names
        self.reset()

    def reset(self):
        self.num_inst, self.sum_metric = 0, 0.0

    def get(self):
        return self.name, (float('nan') if self.num_inst == 0 else self.sum_metric / self.num_inst)

    def update(self, labels, preds):
        raise NotImplementedError


class ScalarLoss(EvalMetric):
    """Running mean of a scalar network output (a loss head): every update adds the sum of preds[0] and counts one instance."""

    def update(self, labels, preds):
        v = preds[0]
        v = v.asnumpy() if hasattr(v, "asnumpy") else np.asarray(v.cpu() if hasattr(v, "cpu") else v)
        self.sum_metric += float(np.sum(v))
        self.num_inst += 1
