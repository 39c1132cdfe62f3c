"""ORACLE — build-container-only helper: import the REAL reference (`/root/reference`)
on PyTorch-CPU.

The reference's `model/x3d.py` imports `fvcore` / `pytorchvideo`, which are neither
vendored nor installed.  We inject this repo's restatement of those 11
symbols (`oracle/pv.py`) into `sys.modules` before importing, so that every line of the
reference's own `model/x3d.py`, `model/trainer.py`, `model/change_decoder.py`,
`model/utils.py` executes unmodified.  Used only by `oracle/gen_golden.py` and the
CPU test that cross-checks the restatement when `/root/reference` exists; nothing on the
GPU box imports this (the reference does not travel).
"""
import os
import sys
import types

from . import pv

REFERENCE_ROOT = "/root/reference"


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "model", "x3d.py"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    """Returns the reference's `model.trainer`, `model.utils`, `utils.metric_tool` modules."""
    if not reference_available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference
    _mod("fvcore"); _mod("fvcore.nn")
    _mod("fvcore.nn.squeeze_excitation", SqueezeExcitation=pv.SqueezeExcitation)
    _mod("pytorchvideo"); _mod("pytorchvideo.layers"); _mod("pytorchvideo.models")
    _mod("pytorchvideo.layers.convolutions", Conv2plus1d=pv.Conv2plus1d)
    _mod("pytorchvideo.layers.swish", Swish=pv.Swish)
    _mod("pytorchvideo.layers.utils", round_repeats=pv.round_repeats, round_width=pv.round_width,
         set_attributes=pv.set_attributes)
    _mod("pytorchvideo.models.head", ResNetBasicHead=pv.ResNetBasicHead)
    _mod("pytorchvideo.models.net", Net=pv.Net)
    _mod("pytorchvideo.models.resnet", BottleneckBlock=pv.BottleneckBlock, ResBlock=pv.ResBlock,
         ResStage=pv.ResStage)
    _mod("pytorchvideo.models.stem", ResNetBasicStem=pv.ResNetBasicStem)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # The reference's top-level packages are called `model` / `utils`; make sure no
    # same-named module from elsewhere shadows them.
    for name in [n for n in sys.modules if n == "model" or n.startswith("model.")
                 or n == "utils" or n.startswith("utils.")]:
        del sys.modules[name]
    import importlib
    trainer = importlib.import_module("model.trainer")
    mutils = importlib.import_module("model.utils")
    metric = importlib.import_module("utils.metric_tool")
    return trainer, mutils, metric
