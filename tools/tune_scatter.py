#!/usr/bin/env python3
"""A/B timing of the packet decode kernels in ONE process, interleaved rounds (guide rule 24).

Variants are selected through the library's environment knobs, set per context with ldpc_amd_configure:
    LDPC_AMD_APPLY=gather|scatter   LDPC_AMD_SCATTER_R=2|4|8   LDPC_AMD_SCATTER_NT=0|1   LDPC_AMD_SCATTER_THREADS=...
Prints median / min device time (HIP events inside the library) of the peel and apply kernels per variant.
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--S", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--per", type=float, default=0.10)
    ap.add_argument("--variants", type=str, default="")
    args = ap.parse_args()
    import torch
    from ldpc_erasure_codes_amd import api, codes

    dev = torch.device("cuda", 0)
    ctx = api.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    n, k, _ = ctx.code_info(h)
    F, S = args.frames, args.S
    src = torch.empty((F, k, S), dtype=torch.uint8, device=dev)
    ctx.synth_source(1, 0, F, k, S, src)
    cw = ctx.encode(h, src)
    del src
    era = torch.empty((F, n), dtype=torch.uint8, device=dev)
    ctx.synth_erasures_uniform(2, 0, F, n, args.per, era)
    sym = cw.clone()
    sym[era.bool()] = 0x5A
    out = torch.empty_like(sym)
    sw = torch.empty(F, dtype=torch.int32, device=dev)
    res = torch.empty_like(sw)
    st = torch.empty_like(sw)

    variants = {
        "gather": {"LDPC_AMD_APPLY": "gather"},
        "scatter 1tier": {"LDPC_AMD_SCATTER_TIERS": "1", "LDPC_AMD_SCATTER_NT": "0"},
        "scatter 1tier nt": {"LDPC_AMD_SCATTER_TIERS": "1", "LDPC_AMD_SCATTER_NT": "1"},
        "scatter 2tier": {"LDPC_AMD_SCATTER_NT": "0"},
        "scatter 2tier nt": {"LDPC_AMD_SCATTER_NT": "1"},
        "scatter 2tier nt R1": {"LDPC_AMD_SCATTER_NT": "1", "LDPC_AMD_SCATTER_R": "1"},
        "scatter 2tier nt R2": {"LDPC_AMD_SCATTER_NT": "1", "LDPC_AMD_SCATTER_R": "2"},
        "scatter 2tier nt R4": {"LDPC_AMD_SCATTER_NT": "1", "LDPC_AMD_SCATTER_R": "4"},
        "xcd0": {"LDPC_AMD_SCATTER_XCD": "0"},
        "B128": {"LDPC_AMD_SCATTER_B": "128"},
        "B128 R4": {"LDPC_AMD_SCATTER_B": "128", "LDPC_AMD_SCATTER_R": "4"},
        "xcd1": {"LDPC_AMD_SCATTER_XCD": "1"},
        "dyn0": {"LDPC_AMD_SCATTER_DYN": "0"},
        "dyn1": {"LDPC_AMD_SCATTER_DYN": "1"},
        "dyn2 list": {"LDPC_AMD_SCATTER_DYN": "2"},
        "dyn3 sorted list": {"LDPC_AMD_SCATTER_DYN": "3"},
        "dyn4 windowed sorted list": {"LDPC_AMD_SCATTER_DYN": "4"},
    }
    if args.variants:
        variants = {k: v for k, v in variants.items() if any(x in k for x in args.variants.split(","))}
    knobs = ["LDPC_AMD_APPLY", "LDPC_AMD_SCATTER_TIERS", "LDPC_AMD_SCATTER_NT", "LDPC_AMD_SCATTER_R", "LDPC_AMD_SCATTER_XCD", "LDPC_AMD_SCATTER_B", "LDPC_AMD_SCATTER_DYN"]
    times = {name: {"peel": [], "apply": []} for name in variants}
    ctx.set_profiling(True)
    for rnd in range(args.rounds + 1):
        for name, env in variants.items():
            for kk in knobs:
                ctx.configure(kk, None)
            ctx.configure_many(env)
            ctx.decode(h, sym, era, out=out, sweeps=sw, residual=res, status=st)
            prof = ctx.get_profile()
            if rnd == 0:
                assert torch.equal(out, cw), name  # every variant must stay bit-exact
                continue
            times[name]["peel"].append(prof["peel"][0])
            times[name]["apply"].append(prof["apply"][0])
    alg = (2 * n * S + n + 8) * F
    for name, t in times.items():
        a = statistics.median(t["apply"])
        print(f"{name:22s} apply median {a:7.3f} ms (min {min(t['apply']):7.3f})  peel {statistics.median(t['peel']):6.3f} ms"
              f"  -> {alg / a / 1e6:7.1f} GB/s algorithmic, frac {alg / a / 1e6 / 8000:.3f}")
    ctx.close()


if __name__ == "__main__":
    main()
