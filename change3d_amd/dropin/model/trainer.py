"""reference `model/trainer.py` import path -> change3d_amd.model.trainer (see ../README.md)."""
from change3d_amd.model.trainer import *  # noqa: F401,F403
from change3d_amd.model import trainer as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
globals().update({n: getattr(_impl, n) for n in __all__})
