#!/usr/bin/env python3
"""Diagnostic: where does the ML fast path (ldpc_ml_pi_kernel, one wavefront per residual frame) spend its time?  Builds a SEPARATE
library with -DLDPC_AMD_STAMPS and prints the share of each phase over the residual frames of the BASELINE cfg 3 batch (the fast
path looks at the erasure pattern only, so S = 16 is used).  Never quote this build's run time."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = {40: "unknowns: compact indices", 41: "checks: counts, XOR sums, first queue", 42: "peel + inactivate (+ symbolic pushes)", 43: "candidate scan",
          45: "Gauss-Jordan", 46: "schedule: count pass", 47: "schedule: place pass"}


def main():
    so = "/tmp/libldpc_erasure_amd_stamps.so"
    src = os.path.join(ROOT, "ldpc_erasure_codes_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-strict-aliasing",
                           "-DLDPC_AMD_STAMPS", "-shared", "-o", so, os.path.join(src, "kernels.hip"), os.path.join(src, "api.cpp"), os.path.join(src, "wire.cpp")])
    import torch
    from ldpc_erasure_codes_amd import api, codes, synth
    api.LIB_PATH = so
    ctx = api.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.configure("ML_PI", "1")
    L = api.load_library()
    L.ldpc_amd_debug_peel_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    n, k, _ = ctx.code_info(h)
    dev = torch.device("cuda", 0)
    era_np = synth.erasures_bursty(31, 0, 4096, n, 0.13, 0.8, 10.0)
    era_np = np.ascontiguousarray(era_np[era_np.sum(axis=1) < n - k])
    F = era_np.shape[0]
    era = torch.from_numpy(era_np).to(dev)
    S = 16
    src_t = torch.empty((F, k, S), dtype=torch.uint8, device=dev)
    ctx.synth_source(11, 0, F, k, S, src_t)
    cw = ctx.encode(h, src_t).reshape(F, n, S)
    buf = (C.c_ulonglong * 56)()
    out, sw, rs, st = ctx.decode(h, cw, era)
    ctx.synchronize()
    L.ldpc_amd_debug_peel_stamps(ctx._h, buf, 1)
    reps = 3
    for _ in range(reps):
        ctx.decode(h, cw, era)
    L.ldpc_amd_debug_peel_stamps(ctx._h, buf, 1)
    nml = int((st.cpu().numpy() >= 1).sum())
    tot = sum(buf[i] for i in PHASES)
    print(f"{nml} residual frames; {tot / reps / max(nml, 1):.0f} ticks per frame (s_memtime, 100 MHz)")
    for i, name in PHASES.items():
        print(f"   {name:42s} {buf[i] / reps / max(nml, 1):10.0f} ticks per frame  {100.0 * buf[i] / max(tot, 1):5.1f} %")
    # the solve kernel on the same batch at S = 1024 (wavefront 0 of every workgroup)
    SOLVE = {32: "solve: task pick-up, table, zeroing", 33: "solve: level 0 (known neighbours from HBM)", 34: "solve: the levels in between",
             36: "solve: output level + wait"}
    del cw, src_t, out
    S = 1024
    src_t = torch.empty((F, k, S), dtype=torch.uint8, device=dev)
    ctx.synth_source(11, 0, F, k, S, src_t)
    cw = ctx.encode(h, src_t).reshape(F, n, S)
    del src_t
    out = torch.empty_like(cw)
    stt = torch.empty(F, dtype=torch.int32, device=dev)
    for mode in ("1", "2", "0"):
        ctx.configure("ML_PI", mode)
        ctx.decode(h, cw, era, out=out, status=stt)
        L.ldpc_amd_debug_peel_stamps(ctx._h, buf, 1)
        ctx.decode(h, cw, era, out=out, status=stt)
        L.ldpc_amd_debug_peel_stamps(ctx._h, buf, 1)
        stot = sum(buf[i] for i in SOLVE)
        print(f"ML_PI={mode}, S=1024: solve kernel, {stot / max(nml, 1):.0f} ticks per frame (all slices, wavefront 0)")
        for i, name in SOLVE.items():
            print(f"      {name:46s} {buf[i] / max(nml, 1):10.0f} ticks  {100.0 * buf[i] / max(stot, 1):5.1f} %")
    ctx.close()


if __name__ == "__main__":
    main()
