"""Host-side binding of libdasac_hip.so (hand-written gfx950 kernels behind a C ABI).

`lib`  -- ctypes loader (fails loudly when the library is missing: there is no CPU fallback)
`ops`  -- tensor-level wrappers + autograd Functions used by the drop-in `models` package
"""
from .lib import load, LIB_PATH, DasacError  # noqa: F401
