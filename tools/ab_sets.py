#!/usr/bin/env python3
"""Same-process A/B of knob SETS on the cfg 2 / cfg 3 packet batches: interleaved rounds, per-kind HIP-event times and the wall time of
a step (HIP events around three calls), same bytes required.
    python tools/ab_sets.py "" "ML_PI_LDS=160,ML_PI_WAVES=4,ML_PI_WGS=24" [--cfg 3] [--rounds 6]"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sets", nargs="+")
    ap.add_argument("--cfg", nargs="+", default=["3"])
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--S", type=int, default=1024)
    ap.add_argument("--drop-undecoded", action="store_true", help="diagnostic: leave out the frames the decoder cannot finish (rank-deficient residual systems)")
    args = ap.parse_args()
    import torch
    import bench

    class A:
        pass
    g = bench.Gpu(A(), 0, 1, 0)
    ctx = g.ctx
    sets = [[kv.split("=") for kv in s.split(",") if kv] for s in args.sets]
    for cfg in args.cfg:
        code_ind = 3 if cfg == "4p" else 1
        h, n, k = g.code(code_ind)
        cw, sym, era, _ = g.make_batch("cfg" + cfg, code_ind, args.S, frame0=0, nframes=4096)
        out = torch.empty_like(sym)
        st = torch.empty(sym.shape[0], dtype=torch.int32, device=g.dev)
        if args.drop_undecoded:
            ctx.decode(h, sym, era, out=out, status=st)
            torch.cuda.synchronize()
            keep = st <= 1
            cw, sym, era = cw[keep].contiguous(), sym[keep].contiguous(), era[keep].contiguous()
            out = torch.empty_like(sym)
            st = torch.empty(sym.shape[0], dtype=torch.int32, device=g.dev)
            print(f"cfg{cfg}: {int(keep.sum())} of {keep.numel()} frames kept", flush=True)
        times = [{"peel": [], "apply": [], "ml": [], "wall": []} for _ in sets]
        ref = None
        for rnd in range(args.rounds + 1):
            for i, s in enumerate(sets):
                for kk, v in s:
                    ctx.configure(kk, v)
                out.fill_(0xEE)
                ctx.decode(h, sym, era, out=out, status=st)
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    ctx.decode(h, sym, era, out=out, status=st)
                e1.record(); torch.cuda.synchronize()
                wall = e0.elapsed_time(e1) / 3
                ctx.get_profile(); ctx.set_profiling(True)
                for _ in range(3):
                    ctx.decode(h, sym, era, out=out, status=st)
                ctx.set_profiling(False)
                p = ctx.get_profile()
                for kk, v in s:
                    ctx.configure(kk, None)
                if rnd == 0:
                    if ref is None:
                        ref = out.clone()
                    else:
                        assert torch.equal(ref, out), s
                    continue
                for kd in ("peel", "apply", "ml"):
                    times[i][kd].append(p[kd][0] / 3)
                times[i]["wall"].append(wall)
        for i, s in enumerate(sets):
            med = {kd: statistics.median(t) for kd, t in times[i].items()}
            print(f"cfg{cfg} S={args.S} [{args.sets[i] or 'defaults'}]: peel {med['peel']:.3f}  apply {med['apply']:.3f}  ml {med['ml']:.3f}  sum {med['peel'] + med['apply'] + med['ml']:.3f}  wall {med['wall']:.3f} ms", flush=True)
        del cw, sym, era, out, ref
        torch.cuda.empty_cache()
    g.close()


if __name__ == "__main__":
    main()
