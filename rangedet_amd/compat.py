"""``install_aliases()`` registers this package's mirrors under the reference's import paths so that reference-style
model/config code (``from mxnext.complicate import normalizer_factory``, ``from rangedet.symbol.backbone.dla_backbone
import DLABackbone``, ``from processing_cxx import wnms_4c`` ...) resolves to the HIP-backed implementation."""
import importlib
import sys
import types

_MAP = {
    "mxnext": "rangedet_amd.mxnext",
    "mxnext.simple": "rangedet_amd.mxnext.simple",
    "mxnext.complicate": "rangedet_amd.mxnext.complicate",
    "rangedet.symbol.backbone.meta_kernel": "rangedet_amd.symbol.backbone.meta_kernel",
    "rangedet.symbol.backbone.dla_backbone": "rangedet_amd.symbol.backbone.dla_backbone",
    "rangedet.symbol.head.builder": "rangedet_amd.symbol.head.builder",
    "rangedet.core.input": "rangedet_amd.core.input",
    "rangedet.core.detection_metric": "rangedet_amd.core.detection_metric",
    "processing_cxx": "rangedet_amd.processing_cxx",
}


def install_aliases(include_mxnet=True):
    for pkg in ("rangedet", "rangedet.symbol", "rangedet.symbol.backbone", "rangedet.symbol.head", "rangedet.core"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
    for alias, target in _MAP.items():
        sys.modules[alias] = importlib.import_module(target)
    if include_mxnet and "mxnet" not in sys.modules:
        sys.modules["mxnet"] = importlib.import_module("rangedet_amd.mx")
    return sorted(_MAP)
