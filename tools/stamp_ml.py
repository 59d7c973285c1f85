#!/usr/bin/env python3
"""Diagnostic: where does the ML kernel (wavefront 0 of a workgroup) spend its time?  Builds a SEPARATE library with
-DLDPC_AMD_STAMPS and prints the share of each phase on single frames of the BASELINE cfg 3 batch.  Never quote this
build's run time."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = {16: "build", 17: "barrier wait", 18: "step header", 19: "row updates", 20: "next-column scan",
          21: "packets: backward levels (one wavefront)", 22: "packets: schedule count pass", 23: "packets: schedule place pass",
          24: "fall-back: forward level sort", 25: "fall-back: forward pull", 26: "fall-back: backward levels + sort",
          27: "back substitution / fall-back backward pull", 28: "write-back"}
SOLVE = {32: "solve: task pick-up, table, zeroing", 33: "solve: level 0 (known neighbours from HBM)", 34: "solve: forward + backward levels",
         36: "solve: output level + wait"}


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
    L = api.load_library()
    L.ldpc_amd_debug_peel_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    n, k, _ = ctx.code_info(h)
    dev = torch.device("cuda", 0)
    era_np = synth.erasures_bursty(31, 0, 4096, n, 0.13, 0.8, 10.0)
    era_np = np.ascontiguousarray(era_np[era_np.sum(axis=1) < n - k])
    F = era_np.shape[0]
    era = torch.from_numpy(era_np).to(dev)
    for S in ([int(x) for x in sys.argv[1:]] or [1, 1024]):
        src_t = torch.empty((F, k, S), dtype=torch.uint8, device=dev)
        ctx.synth_source(11, 0, F, k, S, src_t)
        cw = ctx.encode(h, src_t if S > 1 else src_t.reshape(F, k)).reshape(F, n, S)
        del src_t
        out, sw, rs, st = ctx.decode(h, cw, era)
        ctx.synchronize()
        rs_c = rs.cpu().numpy()
        order = np.argsort(rs_c)
        buf = (C.c_ulonglong * 56)()
        for E in (100, 200, 300, 400):
            j = int(order[np.searchsorted(rs_c[order], E, side="left")])
            s1, e1 = cw[j:j + 1].contiguous(), era[j:j + 1].contiguous()
            ctx.decode(h, s1, e1)
            L.ldpc_amd_debug_peel_stamps(ctx._h, buf, 1)
            reps = 3
            for _ in range(reps):
                ctx.decode(h, s1, e1)
            L.ldpc_amd_debug_peel_stamps(ctx._h, buf, 1)
            tot = sum(buf[i] for i in PHASES)
            print(f"S={S} residual {rs_c[j]}: {tot / reps:.0f} ticks per frame; per column {tot / reps / max(rs_c[j], 1):.1f}")
            for i, name in PHASES.items():
                if buf[i]:
                    print(f"   {name:22s} {buf[i] / reps:10.0f} ticks  {100.0 * buf[i] / tot:5.1f} %")
            stot = sum(buf[i] for i in SOLVE)
            if stot:
                print(f"   solve kernel, all slices of the frame, wavefront 0: {stot / reps:.0f} ticks")
                for i, name in SOLVE.items():
                    print(f"      {name:46s} {buf[i] / reps:10.0f} ticks  {100.0 * buf[i] / stot:5.1f} %")
        del cw
    ctx.close()


if __name__ == "__main__":
    main()
