#!/usr/bin/env python3
"""PCIe-inclusive rate of the boundary when the caller hands over HOST buffers (pageable numpy arrays, as the reference's
C host and a MEX gateway do): frames/s of ldpc_amd_decode_batch without LDPC_AMD_DEVICE_PTRS.  This is never the
`value` of bench.py; DESIGN.md quotes it next to the device-resident rate."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from ldpc_erasure_codes_amd import api, codes, synth
    if len(sys.argv) > 1:          # another build of the library (A/B on one box)
        api.LIB_PATH = os.path.abspath(sys.argv[1])
    print("library:", api.LIB_PATH)
    ctx = api.Context(0)
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    n, k, _ = ctx.code_info(h)
    for S, F in ((1024, 1024), (1, 65536)):
        dev = torch.device("cuda", 0)
        src = torch.empty((F, k, S), dtype=torch.uint8, device=dev)
        ctx.synth_source(1, 0, F, k, S, src)
        cw = ctx.encode(h, src if S > 1 else src.reshape(F, k))
        ctx.synchronize()
        cw_h = cw.cpu().numpy().reshape(F, n, S)
        del src, cw
        era = synth.erasures_uniform(2, 0, F, n, 0.10)
        sym = cw_h.copy()
        sym[era.astype(bool)] = 0x5A
        out = np.empty_like(sym)
        sw = np.empty(F, dtype=np.int32)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            ctx.decode(h, sym if S > 1 else sym[:, :, 0], era, out=out if S > 1 else out[:, :, 0], sweeps=sw)
            ts.append(time.perf_counter() - t0)
        assert np.array_equal(out, cw_h)
        t = min(ts)
        gb = 2.0 * F * n * S / 1e9
        print(f"S={S:5d}: {F} frames from/to host memory in {t * 1e3:8.1f} ms -> {F / t:10.0f} frames/s, {gb / t:6.2f} GB/s over PCIe (both directions summed)")
        if S > 1:   # the encoder and the RS decoder through the same pipeline (round 3)
            srch = np.ascontiguousarray(cw_h[:, :k, :])
            ts = []
            for _ in range(2):
                t0 = time.perf_counter()
                enc = ctx.encode(h, srch)
                ts.append(time.perf_counter() - t0)
            assert np.array_equal(enc, cw_h)
            t = min(ts)
            print(f"S={S:5d}: encode of {F} frames from/to host memory in {t * 1e3:8.1f} ms -> {F / t:10.0f} frames/s, {F * (n + k) * S / 1e9 / t:6.2f} GB/s over PCIe")
            for pipe in ((0, 1) if hasattr(ctx._L, "ldpc_amd_configure") else ()):
                ctx.configure("HOST_PIPELINE", pipe)
                t0 = time.perf_counter()
                enc = ctx.encode(h, srch)
                print(f"          encode, HOST_PIPELINE={pipe}: {(time.perf_counter() - t0) * 1e3:8.1f} ms")
            if hasattr(ctx._L, "ldpc_amd_configure"):
                ctx.configure("HOST_PIPELINE", None)
    ctx.close()


if __name__ == "__main__":
    main()
