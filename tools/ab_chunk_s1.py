#!/usr/bin/env python3
"""Same-process A/B of CHUNK_S1 (frames per launch of a long S = 1 batch): wall clock per decode call, interleaved rounds, the same
bytes and status words required.  Batches: cfg 4's (4080,3060) stream, cfg 5's two codes (32768 frames each), (2040,1530) 65536 frames.
    python tools/ab_chunk_s1.py [--values 16384 32768 65536] [--rounds 5]"""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--values", nargs="+", default=["16384", "32768", "65536"])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import bench

    class A:
        pass
    g = bench.Gpu(A(), 0, 1, 0)
    ctx = g.ctx
    for cfg, code_ind, F in (("cfg4", 3, 65536), ("cfg5", 2, 32768), ("cfg5", 1, 32768), ("cfg2", 1, 65536), ("cfg3", 1, 65536)):
        h, n, k = g.code(code_ind)
        cw, sym, era, _ = g.make_batch(cfg, code_ind, 1, frame0=0, nframes=F)
        F = sym.shape[0]
        out = torch.empty_like(sym)
        sw, res, st = (torch.empty(F, dtype=torch.int32, device=g.dev) for _ in range(3))
        times = {v: [] for v in args.values}
        ref = None
        for rnd in range(args.rounds + 1):
            for v in args.values:
                ctx.configure("CHUNK_S1", v)
                out.fill_(0xEE)
                ctx.decode(h, sym, era, out=out, sweeps=sw, residual=res, status=st)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    ctx.decode(h, sym, era, out=out, sweeps=sw, residual=res, status=st)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / args.steps * 1e3
                ctx.configure("CHUNK_S1", None)
                cur = (out.clone(), sw.clone(), res.clone(), st.clone())
                if ref is None:
                    ref = cur
                else:
                    assert all(torch.equal(a, b) for a, b in zip(ref, cur)), v
                if rnd:
                    times[v].append(dt)
        ok = bool(torch.equal(ref[0][ref[3] <= 1], cw[ref[3] <= 1]))
        for v in args.values:
            med = statistics.median(times[v])
            print(f"{cfg} code {code_ind} (n={n}) S=1 {F} frames CHUNK_S1={v}: {med:.3f} ms per call = {F / med / 1e3:.2f} M frames/s  "
                  f"(min {min(times[v]):.3f})  decoded = codewords: {ok}", flush=True)
        del cw, sym, era, out, ref, cur
        torch.cuda.empty_cache()
    g.close()


if __name__ == "__main__":
    main()
