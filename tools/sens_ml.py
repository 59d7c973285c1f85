#!/usr/bin/env python3
"""Diagnostic (-DLDPC_AMD_MLDBG build): sensitivity of the ML factorisation to its parts, S = 1 on the cfg 3 batch.  LDPC_AMD_ML_DBG bits:
1 untouched-row scan twice, 2 a second barrier per column, 4 row update twice, 8 four more dependent LDS round trips on the row-update
chain, 64 the two same-address atomics twice, 128 four more round trips on the bookkeeping thread, 256 four more in every wavefront's
header, 16 no column at all, 32 stop after half of the columns, 4096 E extra empty steps (barrier + header read), 8192 E extra steps with a
row update by every lane group; solve kernel (S = 1024): 512 / 1024 the multiply-accumulates of
level 0 / of the later levels three times, 2048 a second barrier per level.  Never quote this build's run time."""
import os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/libldpc_erasure_amd_mldbg.so"
src = os.path.join(ROOT, "ldpc_erasure_codes_amd", "csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-strict-aliasing", "-DLDPC_AMD_MLDBG",
                       "-shared", "-o", so, os.path.join(src, "kernels.hip"), os.path.join(src, "api.cpp"), os.path.join(src, "wire.cpp")])
import torch
from ldpc_erasure_codes_amd import api
api.LIB_PATH = so
import bench
class A: pass
g = bench.Gpu(A(), 0, 1, 0)
h, n, k = g.code(1)
cw, sym, era, _ = g.make_batch("cfg3", 1, 1, frame0=0, nframes=4096)
out = torch.empty_like(sym)
res = {}
for rnd in range(5):
    for dbg in (0, 1, 2, 4, 8, 64, 128, 256, 16, 32, 4096, 8192):
        g.ctx.configure("LDPC_AMD_ML_DBG", dbg)
        g.ctx.get_profile(); g.ctx.set_profiling(True)
        g.ctx.decode(h, sym, era, out=out)
        g.ctx.set_profiling(False)
        t = g.ctx.get_profile()["ml"][0]
        if rnd: res.setdefault(dbg, []).append(t)
for dbg, v in res.items():
    print(f"dbg {dbg}: ML kernel {statistics.median(v):.3f} ms")
# the solve kernel (packets): 512 level-0 multiply-accumulates three times, 1024 the later levels' three times, 2048 a second barrier
# per level (the ML stage's time is factor + solve; the factor part does not change with these bits)
del cw, sym, era, out
cw, sym, era, _ = g.make_batch("cfg3", 1, 1024, frame0=0, nframes=4096)
out = torch.empty_like(sym)
res = {}
for rnd in range(4):
    for dbg in (0, 512, 1024, 2048):
        g.ctx.configure("LDPC_AMD_ML_DBG", dbg)
        g.ctx.get_profile(); g.ctx.set_profiling(True)
        g.ctx.decode(h, sym, era, out=out)
        g.ctx.set_profiling(False)
        t = g.ctx.get_profile()["ml"][0]
        if rnd: res.setdefault(dbg, []).append(t)
for dbg, v in res.items():
    print(f"S=1024 dbg {dbg}: ML stage (factor + solve) {statistics.median(v):.3f} ms")
