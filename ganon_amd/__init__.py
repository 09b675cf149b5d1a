"""ganon_amd -- MI355X-native implementation of ganon's read-classification hot path.

The product is native: ``csrc/libganon_hip.so`` (hand-written gfx950 HIP kernels behind the C ABI of
``include/ganon_hip.h``) and ``host/ganon-classify`` (C++ drop-in for the reference binary).  This
Python package is only a thin ctypes mirror of the C ABI for tests and benchmarks; importing it never
falls back to a CPU implementation -- if the HIP library is missing or no GPU is visible, calls raise.
"""
from .hip import (ablate, ablation, set_ablation, FILL_3_OF_8, FILL_3_OF_16, GanonHipError, HipFilter, HipGather, HipReassign, HipStream, MATCH_DTYPE, READ_BIG, READ_OK, READ_SMALL, device_count,
                  device_memory, fill_random_words, library_path, load_library)

__all__ = ["ablate", "ablation", "set_ablation", "FILL_3_OF_8", "FILL_3_OF_16", "GanonHipError", "HipFilter", "HipGather", "HipReassign", "HipStream", "device_memory", "MATCH_DTYPE", "READ_OK", "READ_SMALL", "READ_BIG",
           "device_count", "fill_random_words", "library_path", "load_library"]
