#!/usr/bin/env python3
"""A/B timing of the hybrid ML stage on the BASELINE cfg 3 batch (Gilbert-Elliott erasures) in ONE process.

Variants are the library's environment knobs (set per context with ldpc_amd_configure): LDPC_AMD_ML_THREADS=256|512|1024.
Every variant's output must equal the first one's bit for bit (rank-deficient frames included) and the codeword on
every solved frame.  Prints the median device time of the ML kernel (HIP events inside the library).
"""
import argparse
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, nargs="+", default=[1, 1024])
    ap.add_argument("--rounds", type=int, default=4)
    args = ap.parse_args()
    import torch
    from ldpc_erasure_codes_amd import api, codes, synth

    dev = torch.device("cuda", 0)
    ctx = api.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    n, k, _ = ctx.code_info(h)
    era_np = synth.erasures_bursty(31, 0, 4096, n, 0.13, 0.8, 10.0)
    era_np = np.ascontiguousarray(era_np[era_np.sum(axis=1) < n - k])
    F = era_np.shape[0]
    era = torch.from_numpy(era_np).to(dev)
    variants = {"t256": {"LDPC_AMD_ML_THREADS": "256"}, "t512": {"LDPC_AMD_ML_THREADS": "512"},
                "t1024": {"LDPC_AMD_ML_THREADS": "1024"}}
    for S in args.S:
        src = torch.empty((F, k, S), dtype=torch.uint8, device=dev)
        ctx.synth_source(11, 0, F, k, S, src)
        cw = ctx.encode(h, src if S > 1 else src.reshape(F, k))
        if S == 1:
            cw = cw.reshape(F, n, 1)
        del src
        sym = cw.clone()
        sym[era.bool()] = 0x5A
        out = torch.empty_like(sym)
        st = torch.empty(F, dtype=torch.int32, device=dev)
        rs = torch.empty(F, dtype=torch.int32, device=dev)
        ref = None
        times = {v: [] for v in variants}
        applies = {}
        ctx.set_profiling(True)
        for rnd in range(args.rounds + 1):
            for name, env in variants.items():
                ctx.configure_many(env)
                out.zero_()
                ctx.decode(h, sym, era, out=out, residual=rs, status=st)
                prof = ctx.get_profile()
                if rnd == 0:
                    ok = st <= 1
                    assert torch.equal(out[ok], cw[ok]), name
                    if ref is None:
                        ref = out.clone()
                        print(f"S={S}: {F} frames, ML on {int((rs > 0).sum())}, rank-deficient {int((st == 2).sum())}, "
                              f"mean residual {float(rs[rs > 0].float().mean()):.1f}, max {int(rs.max())}")
                    assert torch.equal(out, ref), name + " differs from the first variant"
                    continue
                times[name].append(prof["ml"][0])
                applies.setdefault(name, []).append(prof["apply"][0])
        ctx.set_profiling(False)
        for name, t in times.items():
            print(f"  S={S:5d} {name:18s} ml median {statistics.median(t):7.3f} ms (min {min(t):7.3f})   packet kernels median "
                  f"{statistics.median(applies[name]):7.3f} ms")
        del cw, sym, out, ref
        torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
