"""path_tracer_amd — MI355X-native hot path of jackdbd/path-tracer behind a C ABI.

csrc/      hand-written HIP (gfx950) kernels + the C ABI (libptgpu.so, include/ptgpu.h)
host/      C++ mirror of the reference's host interface (Scene, Camera, Image, render)
_lib.py    ctypes binding of the C ABI;  gpu.py  GpuContext handle
"""
from ._lib import LIB_PATH, PtGpuError  # noqa: F401

__all__ = ["LIB_PATH", "PtGpuError"]
