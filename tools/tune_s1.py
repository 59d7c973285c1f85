#!/usr/bin/env python3
"""S = 1 (Matlab-exact) kernel: frames resident per CU against run time, for BASELINE cfg 2 (code A, 4096 and 65536 frames) and
cfg 4 (code C, 65536 frames).  Variants are the library's knobs: LDPC_AMD_PEEL_GT (code tables in global memory: more frames
per CU), LDPC_AMD_PEEL_WPB (cap on the frames per workgroup).  Interleaved rounds in one process; outputs must be identical."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ldpc_erasure_codes_amd import api, codes  # noqa: E402

ctx = api.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)
variants = {"default": {}, "tables in LDS": {"LDPC_AMD_PEEL_GT": "0"}, "tables in global": {"LDPC_AMD_PEEL_GT": "1"},
            "wpb 8": {"LDPC_AMD_PEEL_WPB": "8"}, "wpb 12": {"LDPC_AMD_PEEL_WPB": "12"}}
for code_ind, F in ((1, 4096), (1, 65536), (3, 65536)):
    if not codes.have_builtin(code_ind):
        continue
    h = ctx.load_builtin_code(code_ind, codes.DEFAULT_COEF_SEED[code_ind])
    n, k, _ = ctx.code_info(h)
    src = torch.empty((F, k, 1), dtype=torch.uint8, device=dev)
    ctx.synth_source(1, 0, F, k, 1, src)
    cw = ctx.encode(h, src.reshape(F, k))
    era = torch.empty((F, n), dtype=torch.uint8, device=dev)
    ctx.synth_erasures_uniform(2, 0, F, n, 0.10, era)
    out = torch.empty_like(cw)
    times = {v: [] for v in variants}
    plans = {}
    for rnd in range(6):
        for name, env in variants.items():
            ctx.configure_many(env)
            try:
                ctx.get_profile()
                ctx.set_profiling(True)
                ctx.decode(h, cw, era, out=out)
                ctx.set_profiling(False)
                t = ctx.get_profile()["peel"][0]
                plans[name] = ctx.last_plan()
            finally:
                for kk in env:
                    ctx.configure(kk, None)
            assert torch.equal(out, cw), name
            if rnd:
                times[name].append(t)
    print(f"code {code_ind} (n={n}), {F} frames, S=1, 10 % erasures")
    for name in variants:
        p = plans[name]
        t = statistics.median(times[name])
        print(f"  {name:18s} {t:7.3f} ms  {F / t / 1e3:7.2f} M frames/s   frames/CU {p['frames_per_cu']:2d}  (workgroup {p['frames_per_workgroup']:2d} frames, "
              f"{p['peel_lds_bytes']} B LDS, {p['peel_lds_per_frame']} B per frame, tables in {'global' if p['tables_in_global'] else 'LDS'})")
