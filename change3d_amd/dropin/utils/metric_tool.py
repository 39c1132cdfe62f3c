"""reference `utils/metric_tool.py` import path -> change3d_amd.utils.metric_tool (see ../README.md)."""
from change3d_amd.utils import metric_tool as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
globals().update({n: getattr(_impl, n) for n in __all__})
