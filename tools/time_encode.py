#!/usr/bin/env python3
"""Encoder timing: 4096 x 1 KB-packet frames of code A through ldpc_amd_encode_batch, variants selected by the
library's environment knobs, interleaved rounds in one process; every variant must produce the gather kernel's bytes."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ldpc_erasure_codes_amd import api  # noqa: E402

ctx = api.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
CODE = int(sys.argv[1]) if len(sys.argv) > 1 else 1
h = ctx.load_builtin_code(CODE, {1: 2040, 3: 4080}[CODE])
n, k, _ = ctx.code_info(h)
F, S = (4096 if CODE == 1 else 2048), 1024
src = torch.empty((F, k, S), dtype=torch.uint8, device="cuda")
ctx.synth_source(1, 0, F, k, S, src)
variants = {"gather": {"LDPC_AMD_APPLY": "gather"}, "scatter B=256 (1 WG/CU)": {"LDPC_AMD_ENC_B": "256"}, "scatter B=128 (2 WG/CU)": {},
            "B=128, rows by column degree": {"LDPC_AMD_ENC_LIST": "1"}, "B=128 dyn0": {"LDPC_AMD_SCATTER_DYN": "0"},
            "B=128 R=4": {"LDPC_AMD_SCATTER_R": "4"}, "B=128 list mode": {"LDPC_AMD_SCATTER_DYN": "2"},
            "B=128 no XCD placement": {"LDPC_AMD_SCATTER_XCD": "0"},
            "B=128, level lists from global memory": {"LDPC_AMD_ENC_CLIST": "0"}}
times = {v: [] for v in variants}
ref = None
cw = torch.empty((F, n, S), dtype=torch.uint8, device="cuda")
for rnd in range(5):
    for name, env in variants.items():
        ctx.configure_many(env)
        try:
            cw.fill_(0xEE)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                ctx.encode(h, src, out=cw)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / 3
        finally:
            for kk in env:
                ctx.configure(kk, None)
        if rnd == 0:
            if ref is None:
                ref = cw.clone()
            else:
                assert torch.equal(ref, cw), name
            continue
        times[name].append(t)
for name in variants:
    t = statistics.median(times[name])
    print(f"{name:28s} {t * 1e3:7.3f} ms   {(k + n) * S * F / t / 1e12:5.2f} TB/s algorithmic (k S in + n S out) = {(k + n) * S * F / t / 8e12:.3f} of 8 TB/s")
