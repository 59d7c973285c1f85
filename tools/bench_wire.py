#!/usr/bin/env python3
"""Host-side rate of the wire front-end (include/ldpc_erasure_amd_wire.h): packets/s and payload GB/s of the
packetiser and of the two-buffer reassembler on ONE host core, (2040,1530) blocks of 1 KB packets, 10 % loss,
re-ordering window 300.  No GPU involved."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from ldpc_erasure_codes_amd import api
    api.load_library()  # (imports torch once: keep that out of the timings)
    n, k, S, F = 2040, 1530, 1024, 48
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, size=(F, n, S), dtype=np.uint8)
    import ctypes as C
    L = api.load_library()
    packets = np.ones((F * n, 8 + S), dtype=np.uint8)       # touched once: page faults are not what is measured
    t0 = time.perf_counter()
    L.ldpc_amd_fec_packetize(frames.ctypes.data, F, n, S, 1, 0, packets.ctypes.data)
    t_pk = time.perf_counter() - t0
    keep = np.nonzero(rng.random(F * n) >= 0.10)[0]
    order = keep[np.argsort(keep + rng.integers(0, 300, size=keep.size), kind="stable")]
    stream = np.ascontiguousarray(packets[order])
    rx = api.FecRx(n, k, S)
    sym = np.ones((F, n, S), dtype=np.uint8)
    er = np.ones((F, n), dtype=np.uint8)
    blocks = np.zeros(F, dtype=np.int32)
    used_c = C.c_long(0)
    t0 = time.perf_counter()
    nb = L.ldpc_amd_fec_rx_push_many(rx._h, stream.ctypes.data, stream.shape[0], sym.ctypes.data, er.ctypes.data, blocks.ctypes.data, F, C.byref(used_c))
    t_rx = time.perf_counter() - t0
    used, blocks = used_c.value, blocks[:nb]
    print(f"packetiser : {F * n / t_pk / 1e6:6.2f} M packets/s  {F * n * S / t_pk / 1e9:6.2f} GB/s payload")
    print(f"reassembler: {used / t_rx / 1e6:6.2f} M packets/s  {used * S / t_rx / 1e9:6.2f} GB/s payload, {len(blocks)} blocks closed, "
          f"{rx.dropped} packets dropped")


if __name__ == "__main__":
    main()
