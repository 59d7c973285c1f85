#!/usr/bin/env python3
"""Same-process A/B of ONE knob on the cfg 2 / cfg 3 packet batches: interleaved rounds, per-kind HIP-event times, same bytes required.
    python tools/ab_knob.py PEEL_RELAX 1 0 [--cfg 2 3] [--rounds 6]"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("knob")
    ap.add_argument("values", nargs="+")
    ap.add_argument("--cfg", nargs="+", default=["2", "3"])
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--S", type=int, default=1024)
    args = ap.parse_args()
    import torch
    import bench

    class A:
        pass
    g = bench.Gpu(A(), 0, 1, 0)
    ctx = g.ctx
    for cfg in args.cfg:
        code_ind = 3 if cfg == "4p" else 1
        h, n, k = g.code(code_ind)
        cw, sym, era, _ = g.make_batch("cfg" + cfg, code_ind, args.S, frame0=0, nframes=4096)
        out = torch.empty_like(sym)
        st = torch.empty(sym.shape[0], dtype=torch.int32, device=g.dev)
        times = {v: {"peel": [], "apply": [], "ml": []} for v in args.values}
        ref = None
        for rnd in range(args.rounds + 1):
            for v in args.values:
                ctx.configure(args.knob, v)
                out.fill_(0xEE)
                ctx.get_profile(); ctx.set_profiling(True)
                for _ in range(3):
                    ctx.decode(h, sym, era, out=out, status=st)
                ctx.set_profiling(False)
                p = ctx.get_profile()
                ctx.configure(args.knob, None)
                if rnd == 0:
                    if ref is None:
                        ref = out.clone()
                    else:
                        assert torch.equal(ref, out), v
                    continue
                for kd in times[v]:
                    times[v][kd].append(p[kd][0] / 3)
        for v in args.values:
            med = {kd: statistics.median(t) for kd, t in times[v].items()}
            print(f"cfg{cfg} S={args.S} {args.knob}={v}: peel {med['peel']:.3f}  apply {med['apply']:.3f}  ml {med['ml']:.3f}  total {sum(med.values()):.3f} ms")
        del cw, sym, era, out, ref
        torch.cuda.empty_cache()
    g.close()


if __name__ == "__main__":
    main()
