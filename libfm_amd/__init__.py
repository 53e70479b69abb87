"""libfm_amd -- MI355X-native hot path of srendle/libfm (fm_model::predict + fm_SGD and their learners).

The product is libfm_amd/libfmx.so (hand-written gfx950 HIP kernels behind the C-ABI of include/fmx.h).
This package is the thin Python host side used by the tests and bench.py:
  capi     ctypes binding of the C-ABI (no fallback: raises when the library or the GPU is missing)
  learner  mirror of the reference's fm_model / fm_learn_sgd_element interface on top of the C-ABI
  build    hipcc recipe
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
