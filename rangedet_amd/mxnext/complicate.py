"""``mxnext.complicate.normalizer_factory`` (reference mxnext/complicate.py:14-85): returns ``bn(data, name=...)``.

At test time every variant ('local'/'localbn', 'fix'/'fixbn', 'sync') is the same inference-mode BatchNorm with
eps = 1e-5 + 1e-10 and fix_gamma=False (complicate.py:14,38), so they all map to one recorded op.
"""
from .. import mx

__all__ = ["normalizer_factory"]


def normalizer_factory(type="local", ndev=None, eps=1e-5 + 1e-10, mom=0.9):
    if callable(type):
        return type
    if type not in ("local", "localbn", "fix", "fixbn", "sync", "syncbn"):
        raise KeyError("Unknown norm type {}".format(type))

    def bn(data, gamma=None, beta=None, moving_var=None, moving_mean=None, name=None, momentum=mom, lr_mult=1.0,
           wd_mult=1.0):
        return mx.BatchNorm(data=data, gamma=gamma, beta=beta, moving_var=moving_var, moving_mean=moving_mean,
                            name=name or data.name + "_bn", fix_gamma=False, eps=eps, momentum=momentum)

    return bn
