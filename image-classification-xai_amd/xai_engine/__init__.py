"""xai_engine -- MI355X-native saliency-attribution engine (host side).

Host code is Python on PyTorch-ROCm (classifier forward/backward, streams, memory); every
element-wise / reduction step of the hot path runs in libxai_hip.so (hand-written gfx950
HIP kernels behind the C ABI of include/xai_hip.h).  The `util/` package next to this one
re-exports these functions under the reference's module paths and signatures.
"""
from ._lib import XaiHipError, LIB_PATH, load as load_library  # noqa: F401

__all__ = ["XaiHipError", "LIB_PATH", "load_library"]
