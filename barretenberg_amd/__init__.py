"""barretenberg_amd -- MI355X-native BN254 G1 MSM + Fr NTT behind barretenberg's prover API.

The product is libbbgpu.so (hand-written HIP for gfx950 + a C ABI, include/bbgpu.h).  This package holds the sources
(csrc/), the C++ shim that re-exports the reference's own symbols (shim/) and a thin ctypes mirror of the reference
interface used by the tests and bench.  There is no CPU fallback: without the built library or a GPU, calls raise.
"""
from .bbgpu import BbGpu, BbGpuError, NTT_KINDS, build_library, library_path  # noqa: F401
