"""videovector_amd -- MI355X-native (gfx950) implementation of the videovec_embedding training hot
path of eevignesh/videovector.  The product is the C-ABI library (include/videovec.h, built from
videovector_amd/csrc into videovector_amd/lib/libvideovec.so) and the C++ Layer/Solver facade on
top of it; this Python package is the thin host mirror used by the tests and the benchmark.
There is no CPU fallback: loading fails loudly if the HIP library is missing."""
from .engine import BatchRing, Engine, Sampler, StepConfig, VVError, lib_path, load_library  # noqa: F401

__all__ = ["BatchRing", "Engine", "Sampler", "StepConfig", "VVError", "lib_path", "load_library"]
