#!/usr/bin/env python3
"""Launch plans of the packet kernel for the (4080,3060) code at S = 1024: the plan the library picks
(128-byte pieces, two tiers, tier-1 cap 455 accumulators) against the alternatives, interleaved rounds in one process on one
box; every variant must reproduce the codewords."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench

    class A:
        pass
    g = bench.Gpu(A(), 0, 1, 0)
    ctx = g.ctx
    h, n, k = g.code(3)
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    cw, sym, era, _ = g.make_batch("cfg4p", 3, 1024, frame0=0, nframes=F)
    out = torch.empty_like(sym)
    st = torch.empty(F, dtype=torch.int32, device=g.dev)
    variants = {
        "default (B=128, two tiers, R=2)": {},
        "B=128 single tier": {"SCATTER_TIERS": 1},
        "B=128 fixed stride": {"SCATTER_DYN": 0},
        "B=128 list mode": {"SCATTER_DYN": 2},
        "B=128 windowed sorted list": {"SCATTER_DYN": 4},
        "B=128 no XCD placement": {"SCATTER_XCD": 0},
    }
    times = {v: [] for v in variants}
    plans = {}
    for rnd in range(5):
        for name, kn in variants.items():
            ctx.configure_many(kn)
            try:
                out.fill_(0xEE)
                ctx.get_profile(); ctx.set_profiling(True)
                for _ in range(3):
                    ctx.decode(h, sym, era, out=out, status=st)
                ctx.set_profiling(False)
                p = ctx.get_profile()
                plans[name] = (ctx.last_plan(), ctx.profile_kernel_names()["apply"])
            finally:
                ctx.configure_many({kk: None for kk in kn})
            assert torch.equal(out, cw) and int(st.max()) == 0, name
            if rnd:
                times[name].append(p["apply"][0] / 3)
    ab = F * (2 * n * 1024 + n + 8)
    print(f"(4080,3060), S=1024, {F} frames, uniform 10 %: packet kernel(s), ms per batch (median of 4 rounds x 3), fraction of 8 TB/s")
    for name in variants:
        t = statistics.median(times[name])
        pl, kn = plans[name]
        print(f"  {name:34s} {t:7.3f} ms  {ab / t / 1e6 / 8000:.3f}   {kn}  B={pl['packet_bytes_per_workgroup']} tcap={pl['tier1_cap']} tiers={1 + pl['two_tiers']}")
    g.close()


if __name__ == "__main__":
    main()
