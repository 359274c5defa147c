"""sylow_amd: MI355X-native batched BN254 pairing / BLS-verify engine behind sylow's API shape.

Product path only: HIP kernels (csrc/) behind the C ABI (include/sylow_hip.h), a ctypes engine,
and a host-side mirror of the reference's public items (api.py).  Nothing here imports oracle/.
"""
from ._lib import SylowHipError, build, load  # noqa: F401
from .engine import Engine  # noqa: F401
