#!/usr/bin/env python3
"""Does a second context on its own stream raise the throughput of the headline batch (peel kernel of batch i+1 beside the packet
kernel of batch i)?  Two contexts, two copies of the cfg 2 batch; prints ms per batch with one stream and with two."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from ldpc_erasure_codes_amd import api, codes
    import bench

    class A:
        pass
    g = bench.Gpu(A(), 0, 1, 0)
    ctx = g.ctx
    h, n, k = g.code(1)
    cw, sym, era, _ = g.make_batch("cfg2", 1, 1024, frame0=0, nframes=4096)
    out1 = torch.empty_like(sym); st1 = torch.empty(sym.shape[0], dtype=torch.int32, device=g.dev)
    out2 = torch.empty_like(sym); st2 = torch.empty(sym.shape[0], dtype=torch.int32, device=g.dev)
    ctx2 = api.Context(0)
    h2 = ctx2.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    torch.cuda.synchronize()

    def run(two, reps=20):
        for _ in range(2):
            ctx.decode(h, sym, era, out=out1, status=st1); ctx2.decode(h2, sym, era, out=out2, status=st2)
        ctx.synchronize(); ctx2.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            if two and (i & 1):
                ctx2.decode(h2, sym, era, out=out2, status=st2)
            else:
                ctx.decode(h, sym, era, out=out1, status=st1)
        ctx.synchronize(); ctx2.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    for rnd in range(3):
        print(f"round {rnd}: one stream {run(False):.3f} ms per batch, two streams {run(True):.3f} ms per batch", flush=True)
    assert torch.equal(out1, cw) and torch.equal(out2, cw)


if __name__ == "__main__":
    main()
