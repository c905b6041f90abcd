"""ipc_amd -- MI355X-native (gfx950) implementation of the IPC Newton time-step hot path.

The product is `libipcgpu.so` (hand-written HIP kernels + C++ host classes behind the
C ABI of `include/ipcgpu.h`).  This Python package is only a ctypes binding of that ABI
for tests, `bench.py` and `__graft_entry__.py`; it contains no numerical fallback: every
call fails loudly when the HIP library or a GPU is missing.
"""
from .lib import (Context, IpcGpuError, NotPositiveDefinite, lib_path, load_library,  # noqa: F401
                  declared_symbols)
