#!/usr/bin/env python3
"""A/B timing of the whole decode on the BASELINE cfg 3 batch (Gilbert-Elliott erasures, bench.py's generator) in ONE
process, interleaved rounds.  Variants are the library's environment knobs (set per context with ldpc_amd_configure).  Every variant's output
must equal the first one's bit for bit (rank-deficient frames included) and the codeword on every solved frame.
Prints the median device time of every kernel kind (HIP events inside the library)."""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, nargs="+", default=[1024])
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--frames", type=int, default=4096, help="frames drawn from the chain (the BASELINE batch is 4096)")
    ap.add_argument("--variants", type=str, default="", help='JSON: {"name": {"ENV": "value"}, ...}')
    args = ap.parse_args()
    import torch
    import bench
    from ldpc_erasure_codes_amd import api

    class A:
        pass
    g = bench.Gpu(A(), 0, 1, 0)
    ctx = g.ctx
    h, n, k = g.code(1)
    variants = {"default": {}}
    if args.variants:
        variants.update(json.loads(args.variants))
    for S in args.S:
        cw, sym, era, _ = g.make_batch("cfg3", 1, S, frame0=0, nframes=args.frames)
        F = cw.shape[0]
        out = torch.empty_like(sym)
        st = torch.empty(F, dtype=torch.int32, device=g.dev)
        rs = torch.empty(F, dtype=torch.int32, device=g.dev)
        ref = None
        times = {v: {"peel": [], "apply": [], "ml": []} for v in variants}
        for rnd in range(args.rounds + 1):
            for name, env in variants.items():
                for kk, vv in env.items():
                    ctx.configure(kk, vv)
                try:
                    out.fill_(0xEE)
                    ctx.get_profile()
                    ctx.set_profiling(True)
                    ctx.decode(h, sym, era, out=out, residual=rs, status=st)
                    ctx.set_profiling(False)
                    prof = ctx.get_profile()
                finally:
                    for kk in env:
                        ctx.configure(kk, None)
                if rnd == 0 and not name.startswith("diag"):
                    ok = st <= 1
                    assert torch.equal(out[ok], cw[ok]), name
                    if ref is None:
                        ref = (out.clone(), st.clone())
                    else:
                        assert torch.equal(out, ref[0]) and torch.equal(st, ref[1]), f"{name}: output differs from the first variant"
                    continue
                if rnd == 0:
                    continue
                for kd in times[name]:
                    times[name][kd].append(prof[kd][0])
        print(f"S={S} frames={F} ML frames={int((rs > 0).sum())} rank-deficient={int((st == 2).sum())}")
        for name in variants:
            t = times[name]
            med = {kd: statistics.median(t[kd]) for kd in t}
            print(f"  {name:28s} peel {med['peel']:.3f}  apply {med['apply']:.3f}  ml {med['ml']:.3f}  total {sum(med.values()):.3f} ms")
        del cw, sym, era, out, ref
        torch.cuda.empty_cache()
    g.close()


if __name__ == "__main__":
    main()
