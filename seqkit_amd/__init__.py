"""seqkit_amd -- MI355X (gfx950) implementation of SeqKit's per-record hot path.

The product is the native library `libseqkit_hip.so` (HIP kernels behind the C ABI of
include/seqkit_hip.h) and the C++ driver `seqkit-hip`.  This package only binds the
C ABI for tests and benchmarks; it holds no CPU fallback: loading fails loudly when
the library has not been built.
"""
from .capi import Lib, load, SkhError, StatsPartial  # noqa: F401
