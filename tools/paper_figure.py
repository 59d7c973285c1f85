#!/usr/bin/env python3
"""BLER against packet erasure rate for the (2040,1530) code (--code 3: the synthesised (4080,3060) code): the experiment behind the paper's figure
Latex/LDPC_triangular_2040_1530_Perf_vs_RS.png (Matlab/LDPCErasureCodes_MessagePassingAlgSim.m:116,134-245): binary code,
uniform erasures `rand <= PER`, My_LDPC_Erasure_Decoder (message passing, 50 sweeps) next to My_LDPC_HybridML_Erasure_Decoder
(10 sweeps + GF(2) elimination) and the RS(255,192)-equivalent count.  Prints one line per PER."""
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(ctx, torch, per, nframes, seed=1, chunk=250000, code_ind=1):
    from ldpc_erasure_codes_amd import api
    p = api.code_params(code_ind)
    n, k, rs_n, rs_k = p[0], p[1], p[4], p[5]
    m = n - k
    h = ctx.load_builtin_code(code_ind, 0)   # coefficient seed 0: the binary H
    dev = torch.device("cuda", 0)
    mp_err = ml_err = rs_err = skipped = 0
    done = 0
    while done < nframes:
        c = min(chunk, nframes - done)
        era = torch.empty((c, n), dtype=torch.uint8, device=dev)
        ctx.synth_erasures_uniform(seed, done, c, n, per, era)
        sym = torch.zeros((c, n), dtype=torch.uint8, device=dev)       # the all-zero codeword (the code is linear)
        e0 = era.sum(dim=1, dtype=torch.int32)
        call = e0 <= m                                                   # decoders called only if num_erasures <= n-k (:207)
        out, sw, res, st = ctx.decode(h, sym, era, max_sweeps=50, do_ml=0)
        mp_err += int(((res > 0) | ~call).sum())
        out, sw, res, st = ctx.decode(h, sym, era, max_sweeps=10, do_ml=1)
        ml_err += int(((st >= 2) | ~call).sum())
        skipped += int((~call).sum())
        blocks = era[:, : (n // rs_n) * rs_n].reshape(c, n // rs_n, rs_n).sum(dim=2, dtype=torch.int32)
        rs_err += int((blocks > rs_n - rs_k).sum())
        done += c
        del era, sym, out
    return dict(per=per, frames=nframes, mp=mp_err, ml=ml_err, rs=rs_err, rs_blocks=nframes * (n // rs_n), skipped=skipped)


if __name__ == "__main__":
    import argparse
    import torch
    from ldpc_erasure_codes_amd import api
    ap = argparse.ArgumentParser()
    ap.add_argument("--code", type=int, default=1, help="1: (2040,1530), the authors' matrix; 3: (4080,3060), the matrix synthesised by "
                    "tools/hgen.cpp (Latex/LDPC_triangular_4080_3060_Perf_vs_RS.png is the reference's plot for the authors' own)")
    a = ap.parse_args()
    ctx = api.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    scale = 1 if a.code == 1 else 4
    for per, nf in ((0.14, 1000000), (0.16, 1000000), (0.18, 1000000), (0.20, 200000), (0.22, 50000)):
        nf //= scale
        r = run(ctx, torch, per, nf, code_ind=a.code)
        print(f"code {a.code} PER {per:.2f}: {nf} frames  MP BLER {r['mp'] / nf:.3g} ({r['mp']})  MP+ML BLER {r['ml'] / nf:.3g} ({r['ml']}, {r['skipped']} skipped)  "
              f"RS BLER {r['rs'] / r['rs_blocks']:.3g}", flush=True)
    ctx.close()
