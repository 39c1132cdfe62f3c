"""reference `model/caption_decoder.py` import path -> change3d_amd.model.caption_decoder (see ../README.md)."""
from change3d_amd.model.caption_decoder import *  # noqa: F401,F403
from change3d_amd.model import caption_decoder as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
globals().update({n: getattr(_impl, n) for n in __all__})
