"""mumpy_hip — ctypes binding of libmumpy_hip.so (include/mumpy_hip.h) for PyTorch-ROCm tensors.

PyTorch is plumbing here (device memory, streams, graphs); the arithmetic is in the HIP library.
There is NO CPU or pure-torch fallback: importing `ops` without the built library, or calling an
op with a non-CUDA tensor, raises.
"""
from . import ops  # noqa: F401
from .lib import load_library, library_path  # noqa: F401
