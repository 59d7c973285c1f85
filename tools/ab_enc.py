#!/usr/bin/env python3
"""Same-process A/B of knob SETS on the packet encoder: F x 1 KB-packet frames of code A (1) / C (3), interleaved rounds, same bytes required.
    python tools/ab_enc.py "ENC_PERSIST=0" "ENC_PERSIST=1" [--codes 1 3] [--rounds 6]"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sets", nargs="+")
    ap.add_argument("--codes", nargs="+", type=int, default=[1, 3])
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--S", type=int, default=1024)
    args = ap.parse_args()
    import torch
    from ldpc_erasure_codes_amd import api
    ctx = api.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    sets = [[kv.split("=") for kv in s.split(",") if kv] for s in args.sets]
    for code in args.codes:
        h = ctx.load_builtin_code(code, {1: 2040, 3: 4080, 0: 2000, 2: 4000}[code])
        n, k, _ = ctx.code_info(h)
        F, S = (4096 if n <= 2040 else 2048), args.S
        src = torch.empty((F, k, S), dtype=torch.uint8, device="cuda")
        ctx.synth_source(1, 0, F, k, S, src)
        cw = torch.empty((F, n, S), dtype=torch.uint8, device="cuda")
        ref = None
        times = [[] for _ in sets]
        for rnd in range(args.rounds + 1):
            for i, s in enumerate(sets):
                for kk, v in s:
                    ctx.configure(kk, v)
                cw.fill_(0xEE)
                ctx.encode(h, src, out=cw)
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    ctx.encode(h, src, out=cw)
                e1.record(); torch.cuda.synchronize()
                for kk, v in s:
                    ctx.configure(kk, None)
                if rnd == 0:
                    if ref is None:
                        ref = cw.clone()
                    else:
                        assert torch.equal(ref, cw), s
                    continue
                times[i].append(e0.elapsed_time(e1) / 3)
        alg = F * (k + (n - k)) * S   # source rows read once, parity rows written once (the source rows are copied: + k rows written)
        alg_w = F * (k + n) * S
        for i, s in enumerate(sets):
            t = statistics.median(times[i])
            print(f"code {code} ({n},{k}) F={F} S={S} [{args.sets[i] or 'defaults'}]: {t:.3f} ms  ({alg_w / t / 1e6:.0f} GB/s with the copy of the source rows = {alg_w / t / 1e6 / 8000:.3f} of 8 TB/s)", flush=True)
        del src, cw, ref
        torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
