#!/usr/bin/env python3
"""Diagnostic: phases of the packet kernel (wavefront 0 of every workgroup) when it runs the ENCODER's static schedule and when it
decodes the headline batch.  Builds a SEPARATE library with -DLDPC_AMD_STAMPS.  Never quote this build's run time."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = {12: "set-up (tables, lists, zeroing)", 13: "streaming phase", 14: "wait for the slowest wavefront of the stream", 15: "level phase + parity rows out"}


def main():
    so = os.path.join(ROOT, "tools", "bin", "libldpc_erasure_amd_stamps.so")   # prebuilt by tools/build_stamps.sh (travels with gpurun)
    src = os.path.join(ROOT, "ldpc_erasure_codes_amd", "csrc")
    if not os.path.exists(so):
        so = "/tmp/libldpc_erasure_amd_stamps.so"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-strict-aliasing",
                           "-DLDPC_AMD_STAMPS", "-shared", "-o", so, os.path.join(src, "kernels.hip"), os.path.join(src, "api.cpp"), os.path.join(src, "wire.cpp")])
    import torch
    from ldpc_erasure_codes_amd import api, codes
    api.LIB_PATH = so
    ctx = api.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    L = api.load_library()
    L.ldpc_amd_debug_peel_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    dev = torch.device("cuda", 0)
    buf = (C.c_ulonglong * 56)()
    for code_ind in (1, 3):
        h = ctx.load_builtin_code(code_ind, codes.DEFAULT_COEF_SEED[code_ind])
        n, k, _ = ctx.code_info(h)
        F, S = 2048, 1024
        src_t = torch.empty((F, k, S), dtype=torch.uint8, device=dev)
        ctx.synth_source(1, 0, F, k, S, src_t)
        cw = torch.empty((F, n, S), dtype=torch.uint8, device=dev)
        ctx.encode(h, src_t, out=cw)
        L.ldpc_amd_debug_peel_stamps(ctx._h, buf, 1)
        ctx.encode(h, src_t, out=cw)
        L.ldpc_amd_debug_peel_stamps(ctx._h, buf, 1)
        tot = sum(buf[i] for i in PHASES)
        print(f"encoder, code {code_ind} (n={n}, k={k}), {F} frames x {S} B: {tot / F:.0f} cycles per frame (all slices, wavefront 0)")
        for i, name in PHASES.items():
            print(f"   {name:48s} {100.0 * buf[i] / max(tot, 1):5.1f} %")
        era = torch.empty((F, n), dtype=torch.uint8, device=dev)
        ctx.synth_erasures_uniform(2, 0, F, n, 0.10, era)
        out = torch.empty_like(cw)
        ctx.decode(h, cw, era, out=out)
        L.ldpc_amd_debug_peel_stamps(ctx._h, buf, 1)
        ctx.decode(h, cw, era, out=out)
        L.ldpc_amd_debug_peel_stamps(ctx._h, buf, 1)
        tot = sum(buf[i] for i in PHASES)
        print(f"decoder, uniform 10 %: {tot / F:.0f} cycles per frame")
        for i, name in PHASES.items():
            print(f"   {name:48s} {100.0 * buf[i] / max(tot, 1):5.1f} %")
        if code_ind == 1:   # the bursty channel of cfg 3: 79 % of the frames in tier 2 (one workgroup per CU)
            import numpy as np
            from ldpc_erasure_codes_amd import synth
            era_np = synth.erasures_bursty(31, 0, 2 * F, n, 0.13, 0.8, 10.0)
            era_np = np.ascontiguousarray(era_np[era_np.sum(axis=1) < n - k][:F])
            era = torch.from_numpy(era_np).to(dev)
            ctx.configure("LDPC_AMD_ML_PI", "0"); ctx.configure("LDPC_AMD_ML_OVERLAP", "0")
            ctx.decode(h, cw, era, out=out)
            L.ldpc_amd_debug_peel_stamps(ctx._h, buf, 1)
            ctx.decode(h, cw, era, out=out)
            L.ldpc_amd_debug_peel_stamps(ctx._h, buf, 1)
            ctx.configure("LDPC_AMD_ML_PI", None); ctx.configure("LDPC_AMD_ML_OVERLAP", None)
            tot = sum(buf[i] for i in PHASES)
            print(f"decoder, bursty channel of cfg 3 (both tiers): {tot / F:.0f} cycles per frame")
            for i, name in PHASES.items():
                print(f"   {name:48s} {100.0 * buf[i] / max(tot, 1):5.1f} %   ({buf[i] / F:.0f})")
            for knob, val in (("SCATTER_PAIRS", "0"), ("SCATTER_XL", "0"), ("PEEL_RELAX", "0")):   # round 4: what each of the three changes does to the phases
                ctx.configure("LDPC_AMD_ML_PI", "0"); ctx.configure("LDPC_AMD_ML_OVERLAP", "0"); ctx.configure(knob, val)
                ctx.decode(h, cw, era, out=out)
                L.ldpc_amd_debug_peel_stamps(ctx._h, buf, 1)
                ctx.decode(h, cw, era, out=out)
                L.ldpc_amd_debug_peel_stamps(ctx._h, buf, 1)
                ctx.configure(knob, None); ctx.configure("LDPC_AMD_ML_PI", None); ctx.configure("LDPC_AMD_ML_OVERLAP", None)
                tot2 = sum(buf[i] for i in PHASES)
                print(f"   ... with {knob}={val}: {tot2 / F:.0f} cycles per frame: " + ", ".join(f"{buf[i] / F:.0f}" for i in PHASES))
        del src_t, cw, era, out
        torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
