#!/usr/bin/env python3
"""Encoder, level-by-level against the grouped static schedule (round 4): F x 1 KB-packet frames of code A / C, one context per
ENC_CAP (the cap is read when the code is registered), ENC_GROUP 0 / 1, interleaved rounds in one process; same bytes required.
    python tools/time_enc_groups.py [code_ind] [caps...]"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ldpc_erasure_codes_amd import api  # noqa: E402

CODE = int(sys.argv[1]) if len(sys.argv) > 1 else 1
CAPS = [int(x) for x in sys.argv[2:]] or [2, 4, 8, 16, 32]
F, S = (4096 if CODE == 1 else 2048), 1024
ctxs = {}
for cap in CAPS:
    c = api.Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.configure("ENC_CAP", str(cap))
    ctxs[cap] = (c, c.load_builtin_code(CODE, {1: 2040, 3: 4080, 0: 2000, 2: 4000}[CODE]))
c0, h0 = ctxs[CAPS[0]]
n, k, _ = c0.code_info(h0)
src = torch.empty((F, k, S), dtype=torch.uint8, device="cuda")
c0.synth_source(1, 0, F, k, S, src)
cw = torch.empty((F, n, S), dtype=torch.uint8, device="cuda")
variants = [("level by level", CAPS[0], "0")] + [(f"grouped cap={cap}", cap, "1") for cap in CAPS]
times = {v[0]: [] for v in variants}
ref = None
for rnd in range(6):
    for name, cap, grp in variants:
        c, h = ctxs[cap]
        c.configure("ENC_GROUP", grp)
        cw.fill_(0xEE)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            c.encode(h, src, out=cw)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 3
        if rnd == 0:
            if ref is None:
                ref = cw.clone()
            else:
                assert torch.equal(ref, cw), name
            continue
        times[name].append(t)
for name, cap, grp in variants:
    t = statistics.median(times[name])
    info = ctxs[cap][0].encode_info(ctxs[cap][1])
    print(f"code {CODE} {name:18s} {t * 1e3:7.3f} ms  {(k + n) * S * F / t / 8e12:.3f} of 8 TB/s   levels {info['levels']} groups {info['groups']} "
          f"pulls {info['pull_entries']} scatter {info['scatter_entries']} maxpull {info['max_pull']} used {info['last_encode_grouped']}")
