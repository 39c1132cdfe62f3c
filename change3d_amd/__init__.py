"""change3d_amd — MI355X (gfx950) native implementation of the Change3D BCD hot path.

Layout:
  csrc/      hand-written HIP kernels + the C ABI (include/change3d_hip.h)
  lib/       libchange3d_hip.so, built in-tree by __graft_entry__.build()
  _lib.py    ctypes binding (fails loudly when the library is missing — no fallback)
  ops.py     torch-tensor wrappers (device memory + streams only)
  model/     mirror of the reference's model/{x3d,change_decoder,trainer,utils}.py surface
  utils/     mirror of the reference's utils/metric_tool.py
  parallel.py  data-parallel step: one flat-gradient RCCL all-reduce on a side stream
"""
__version__ = "0.1.0"
