"""fastq-rs_amd — MI355X-native FASTQ record scan + per-read statistics behind the `fastq` crate's
Parser / Record / parallel_each surface.

The product is the C-ABI shared library `libfastq_hip.so` (include/fastq_hip.h, built from csrc/ by
hipcc for gfx950) plus the C++ host mirror in host/.  This Python package is only the ctypes stub
the tests and bench.py drive the library through; it has no CPU fallback: importing `binding` fails
loudly when the HIP library has not been built.

The directory name contains '-' (it is the reference's repo name), so load it with
`__graft_entry__.load_package()` (importlib by path) rather than `import`.
"""
from .binding import *  # noqa: F401,F403
from .binding import __all__  # noqa: F401
