#!/usr/bin/env python3
"""Upper bounds for two restructurings of the cfg 3 packet step (VERDICT r2 #1 a / b), measured with the diagnostic build
(-DLDPC_AMD_MLDBG; WRONG bytes in the diagnostic runs -- timing only, never quoted as a result):

  (a) level-split tier 2: the frames with more than tcap steps in two passes of tier-1 geometry (two workgroups per CU).  Bit 32768
      lets the tier-1 launch take those frames cut off at tcap steps and drops the tier-2 launch: that IS the first pass (all rows
      streamed and written once, <= tcap accumulators); the second pass (re-stream the rows the later steps consume) comes on top.
  (c) (round 4) chain composition in the solve schedules: bits 4096 / 8192 leave out the barrier behind all but every second / fourth level of
      the solve kernel -- what a schedule with half / a quarter of the levels could gain at most (the ops stay).
  (e) (round 4) the DECODER's dependency levels collapsed the way the encoder's static schedule is (groups of levels, steps pulling in-group
      accumulators): bit 128 takes the levels of the packet kernels in pairs, bit 64 all steps of a frame as one level -- the most it could gain.
  (b) right-hand sides of the ML systems built inside the packet stream: bit 16384 skips level 0 of the solve kernel (the known-row
      re-read and its multiply-accumulates) -- the most (b) could remove, before the cost of doing the same products in the packet
      kernel and of moving the finished right-hand sides through HBM.

    python tools/bound_cfg3.py [path/to/mldbg.so]
"""
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    pre = os.path.join(ROOT, "tools", "bin", "libldpc_erasure_amd_mldbg.so")   # prebuilt: EXTRA_HIPCC_FLAGS=-DLDPC_AMD_MLDBG LDPC_AMD_OUT=... build.sh
    so = sys.argv[1] if len(sys.argv) > 1 else (pre if os.path.exists(pre) else "/tmp/libldpc_erasure_amd_mldbg.so")
    if len(sys.argv) <= 1 and so != pre:
        src = os.path.join(ROOT, "ldpc_erasure_codes_amd", "csrc")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-strict-aliasing", "-DLDPC_AMD_MLDBG",
                               "-shared", "-o", so, os.path.join(src, "kernels.hip"), os.path.join(src, "api.cpp"), os.path.join(src, "wire.cpp")])
    import torch
    from ldpc_erasure_codes_amd import api
    api.LIB_PATH = os.path.abspath(so)
    import bench

    class A:
        pass
    g = bench.Gpu(A(), 0, 1, 0)
    ctx = g.ctx
    h, n, k = g.code(1)
    cw, sym, era, _ = g.make_batch("cfg3", 1, 1024, frame0=0, nframes=4096)
    out = torch.empty_like(sym)
    st = torch.empty(sym.shape[0], dtype=torch.int32, device=g.dev)
    nst = None
    res = {}
    for rnd in range(5):
        for name, dbg in (("product path", 0), ("(a) tier 1 takes every frame, cut off at tcap steps; no tier 2", 32768), ("(b) solve kernel without level 0", 16384),
                          ("(e) ML_PI=2, packet kernels: the levels of a frame taken in pairs", 128), ("(e) ML_PI=2, packet kernels: every step of a frame as ONE level", 64),
                          ("fast path without its consistency test (ML_PI=2)", -2),
                          ("(c) ML_PI=2, solve kernel: barrier behind every second level only", 4096), ("(c) ML_PI=2, ... behind every fourth level only", 8192),
                          ("(c) ML_PI=2, every forward / backward op as ONE level", 256)):
            # (the (c) runs produce wrong bytes, which the consistency test of the default mode would flag and redo: they run without it,
            # next to a baseline without it)
            ctx.configure("ML_PI", "2" if dbg in (-2, 4096, 8192, 256, 64, 128) else None)
            ctx.configure("ML_DBG", max(dbg, 0))
            ctx.get_profile(); ctx.set_profiling(True)
            ctx.decode(h, sym, era, out=out, status=st)
            ctx.set_profiling(False)
            p = ctx.get_profile()
            if rnd:
                res.setdefault(name, []).append((p["peel"][0], p["apply"][0], p["ml"][0]))
            if dbg == 0 and rnd == 0:
                ok = st <= 1
                assert torch.equal(out[ok], cw[ok])
    ctx.configure("ML_DBG", None)
    ctx.configure("ML_PI", None)
    print(f"cfg 3, S = 1024, {sym.shape[0]} frames; ms per step (median of 4): peel / packet kernels / ML stage (factor + solve)")
    for name, v in res.items():
        print(f"  {name:70s} {statistics.median(x[0] for x in v):.3f} / {statistics.median(x[1] for x in v):.3f} / {statistics.median(x[2] for x in v):.3f}")
    g.close()


if __name__ == "__main__":
    main()
