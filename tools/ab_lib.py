#!/usr/bin/env python3
"""A/B of two BUILDS of the library on one box (box-to-box spread is 3-5 %, more than most kernel changes are worth):
    python tools/ab_lib.py old.so new.so [--rounds 3]
Every round starts one child process per library, alternating; a child times the packet kernel on the cfg 2 batch (plain and
in place), the cfg 3 step at S = 1024 and at S = 1 and the encoder, with the in-library HIP events.  Medians per library are printed."""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(so):
    import torch
    from ldpc_erasure_codes_amd import api
    api.LIB_PATH = so
    import bench

    class A:
        pass
    g = bench.Gpu(A(), 0, 1, 0)
    ctx = g.ctx
    h, n, k = g.code(1)
    res = {}

    def timed(fn, kinds, reps=6):
        fn()
        ctx.get_profile(); ctx.set_profiling(True)
        for _ in range(reps):
            fn()
        ctx.set_profiling(False)
        p = ctx.get_profile()
        return {kd: p[kd][0] / reps for kd in kinds}   # ms per call (a call may launch a kind twice: the two tiers)
    cw, sym, era, _ = g.make_batch("cfg2", 1, 1024, frame0=0, nframes=4096)
    out = torch.empty_like(sym)
    st = torch.empty(sym.shape[0], dtype=torch.int32, device=g.dev)
    t2 = timed(lambda: ctx.decode(h, sym, era, out=out, status=st), ["peel", "apply"])
    res["cfg2_apply"], res["cfg2_peel"] = t2["apply"], t2["peel"]
    assert torch.equal(out, cw)
    src = cw[:, :k, :].contiguous()
    enc = torch.empty_like(cw)
    import time
    ctx.encode(h, src, out=enc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        ctx.encode(h, src, out=enc)
    torch.cuda.synchronize()
    res["encode"] = (time.perf_counter() - t0) / 6 * 1e3
    assert torch.equal(enc, cw)
    del cw, sym, era, out, src, enc
    torch.cuda.empty_cache()
    cw, sym, era, _ = g.make_batch("cfg3", 1, 1024, frame0=0, nframes=4096)
    out = torch.empty_like(sym)
    st = torch.empty(sym.shape[0], dtype=torch.int32, device=g.dev)
    t = timed(lambda: ctx.decode(h, sym, era, out=out, status=st), ["peel", "apply", "ml"])
    res["cfg3_apply"], res["cfg3_ml"], res["cfg3_peel"] = t["apply"], t["ml"], t["peel"]
    ok = st.cpu().numpy() <= 1
    assert torch.equal(out[torch.from_numpy(ok).to(g.dev)], cw[torch.from_numpy(ok).to(g.dev)])
    del cw, sym, era, out
    torch.cuda.empty_cache()
    cw, sym, era, _ = g.make_batch("cfg3", 1, 1, frame0=0, nframes=4096)
    out = torch.empty_like(sym)
    st = torch.empty(sym.shape[0], dtype=torch.int32, device=g.dev)
    res["cfg3_s1_ml"] = timed(lambda: ctx.decode(h, sym, era, out=out, status=st), ["ml"], reps=10)["ml"]
    ok = torch.from_numpy(st.cpu().numpy() <= 1).to(g.dev)
    assert torch.equal(out[ok], cw[ok])
    print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--child", type=str, default="")
    args = ap.parse_args()
    if args.child:
        child(args.child)
        return
    acc = {so: {} for so in args.libs}
    for _ in range(args.rounds):
        for so in args.libs:
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", os.path.abspath(so)], capture_output=True, text=True)
            if o.returncode:
                print(o.stderr[-2000:])
                sys.exit(1)
            r = json.loads(o.stdout.strip().splitlines()[-1])
            for kd, v in r.items():
                acc[so].setdefault(kd, []).append(v)
    for so, d in acc.items():
        print(os.path.basename(so), {kd: round(statistics.median(v), 3) for kd, v in d.items()})


if __name__ == "__main__":
    main()
