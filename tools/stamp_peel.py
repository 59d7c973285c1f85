#!/usr/bin/env python3
"""Diagnostic: where does a peel-kernel wavefront spend its cycles?  Builds a SEPARATE library with
-DLDPC_AMD_STAMPS (s_memtime stamps at the phase boundaries, summed into a buffer nothing else reads) and prints
the share of each phase for the S = 1 (fused) and packet variants.  Never quote this build's run time."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = ["frame load", "peeling sweeps", "status / ML hand-off", "level sort", "schedule write-out", "per-source lists",
          "S=1 apply", "output store"]


def main():
    so = "/tmp/libldpc_erasure_amd_stamps.so"
    src = os.path.join(ROOT, "ldpc_erasure_codes_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-strict-aliasing",
                           "-DLDPC_AMD_STAMPS", "-shared", "-o", so, os.path.join(src, "kernels.hip"), os.path.join(src, "api.cpp"), os.path.join(src, "wire.cpp")])
    import torch
    from ldpc_erasure_codes_amd import api, codes
    api.LIB_PATH = so
    ctx = api.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    L = api.load_library()
    L.ldpc_amd_debug_peel_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    n, k, _ = ctx.code_info(h)
    F = 4096
    dev = torch.device("cuda", 0)
    cases = [(1, "uniform"), (64, "uniform"), (1024, "uniform"), (1024, "cfg3")]
    if len(sys.argv) > 1 and sys.argv[1] == "cfg3":
        cases = [(1024, "cfg3")]
    for S, channel in cases:
        src_t = torch.empty((F, k, S), dtype=torch.uint8, device=dev)
        ctx.synth_source(1, 0, F, k, S, src_t)
        cw = ctx.encode(h, src_t if S > 1 else src_t.reshape(F, k))
        era = torch.empty((F, n), dtype=torch.uint8, device=dev)
        if channel == "uniform":
            ctx.synth_erasures_uniform(2, 0, F, n, 0.10, era)
        else:   # the BASELINE cfg 3 channel of bench.py (frames with E0 >= n-k are simply left undecodable here)
            ctx.synth_erasures_bursty(20261005 + 1, 0, F, n, 0.13, 0.8, 10.0, era)
        print(f"--- S={S} channel={channel}")
        buf = (C.c_ulonglong * 56)()
        L.ldpc_amd_debug_peel_stamps(ctx._h, buf, 1)
        reps = 5
        for _ in range(reps):
            out, sw, res, st = ctx.decode(h, cw, era, do_ml=0)
        L.ldpc_amd_debug_peel_stamps(ctx._h, buf, 1)
        tot = sum(buf[i] for i in range(8))
        print(f"S={S}: {tot / (reps * F):.0f} cycles per frame (sum of phases, s_memtime ticks)")
        for i, name in enumerate(PHASES):
            if buf[i]:
                print(f"   {name:24s} {buf[i] / (reps * F):9.0f} cycles  {100.0 * buf[i] / tot:5.1f} %")
        stot = sum(buf[i] for i in range(12, 16))
        if stot:
            print(f"   packet kernel (per wavefront sums): {stot / (reps * F):.0f} ticks per frame")
            for i, name in zip(range(12, 16), ["set-up", "streaming phase", "wait after stream", "level phase"]):
                print(f"      {name:22s} {100.0 * buf[i] / stot:5.1f} %")
    ctx.close()


if __name__ == "__main__":
    main()
