"""MI355X-native GF(256) LDPC / Reed-Solomon erasure decoding (hot path of chadac8j/LDPC_Erasure_Codes).

The compute path is hand-written HIP for gfx950 behind the C ABI of include/ldpc_erasure_amd.h
(libldpc_erasure_amd.so).  This package is the thin host-side mirror used by tests and the bench:
code tables (codes), synthetic inputs (synth) and the ctypes binding of the C ABI (api).  There is no
CPU fallback: every decode call fails loudly if the HIP library is missing.
"""
from . import codes, synth  # noqa: F401

__all__ = ["codes", "synth"]
