#!/usr/bin/env python3
"""Would the ML factorisation (latency-bound, 80 KB of LDS per workgroup) hide behind the packet kernel (HBM-bound, ~72 KB per
workgroup) if the two ran on separate streams?  Two contexts, one stream each: A = the cfg 2 packet batch (message passing only),
B = the cfg 3 batch at S = 1 (peel + ML kernel, the same factorisation the packet path runs).  Prints the time of A alone, of B
alone and of both enqueued together: 'together' near max(A, B) says the overlap pays, near A + B says the dispatcher serialises
them (each kernel already fills every CU's LDS)."""
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
    cwA, symA, eraA, _ = g.make_batch("cfg2", 1, 1024, frame0=0, nframes=4096)
    outA = torch.empty_like(symA)
    stA = torch.empty(symA.shape[0], dtype=torch.int32, device=g.dev)
    cwB, symB, eraB, _ = g.make_batch("cfg3", 1, 1, frame0=0, nframes=4096)
    outB = torch.empty_like(symB)
    stB = torch.empty(symB.shape[0], dtype=torch.int32, device=g.dev)
    torch.cuda.synchronize()
    ctx2 = api.Context(0)          # its own stream
    h2 = ctx2.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    s1 = torch.cuda.Stream()
    ctx.set_stream(s1.cuda_stream)

    def runA():
        ctx.decode(h, symA, eraA, out=outA, status=stA)

    def runB():
        ctx2.decode(h2, symB, eraB, out=outB, status=stB)

    def sync():
        ctx.synchronize(); ctx2.synchronize()

    def timed(fns, reps=8):
        for f in fns:
            f()
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            for f in fns:
                f()
        sync()
        return (time.perf_counter() - t0) / reps * 1e3
    # knob variants that shrink the LDS share of either side, so that two packet workgroups and a factorisation fit one CU
    variants = [("defaults", {}, {}),
                ("A: 128-byte slices", {"LDPC_AMD_SCATTER_B": "128"}, {}),
                ("B: 4 systems per CU (40 KB each)", {}, {"LDPC_AMD_ML_PACK": "4"}),
                ("A: 128-byte slices, B: 4 per CU", {"LDPC_AMD_SCATTER_B": "128"}, {"LDPC_AMD_ML_PACK": "4"}),
                ("A: 128-byte slices, B: 1 per CU (160 KB)", {"LDPC_AMD_SCATTER_B": "128"}, {"LDPC_AMD_ML_PACK": "1"})]
    for name, ka, kb in variants:
        for kk, v in ka.items():
            ctx.configure(kk, v)
        for kk, v in kb.items():
            ctx2.configure(kk, v)
        for rnd in range(2):
            a, b, ab, ba = timed([runA]), timed([runB]), timed([runA, runB]), timed([runB, runA])
            print(f"{name}, round {rnd}: A alone {a:.3f} ms, B alone {b:.3f} ms, A then B enqueued {ab:.3f} ms, B then A {ba:.3f} ms (sum {a + b:.3f})", flush=True)
        for kk in ka:
            ctx.configure(kk, None)
        for kk in kb:
            ctx2.configure(kk, None)
    assert torch.equal(outA, cwA)
    ok = torch.from_numpy(stB.cpu().numpy() <= 1).to(g.dev)
    assert torch.equal(outB[ok], cwB[ok])
    print("outputs verified")


if __name__ == "__main__":
    main()
