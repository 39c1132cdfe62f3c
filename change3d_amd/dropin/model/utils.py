"""reference `model/utils.py` import path -> change3d_amd.model.utils (see ../README.md)."""
from change3d_amd.model.utils import *  # noqa: F401,F403
from change3d_amd.model import utils as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
globals().update({n: getattr(_impl, n) for n in __all__})
