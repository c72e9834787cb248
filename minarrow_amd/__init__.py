"""minarrow_amd — MI355X (gfx950) kernel layer for Minarrow's src/kernels hot path.

The product is the C ABI in include/minarrow_hip.h, implemented by hand-written HIP kernels in
minarrow_amd/csrc and built into minarrow_amd/lib/libminarrow_hip.so. This package is the thin host side:
`ffi` (ctypes binding of the ABI) and `host` (buffer helpers used by tests and bench.py).
"""
from . import ffi  # noqa: F401

__all__ = ["ffi"]
__version__ = "0.1.0"
