"""Reproduction of the only decoder OUTPUT the reference publishes: Table I of the paper
(Latex/Milcom_2022_ErasureCodes.tex:189-217), block error rates of the FPGA harness
data_in -> ldpc_erasure_decoder -> data_out (OpenCL/host/src/main.cpp:578-626) at raw PER = p/64.

    (2040,1530)  PER 12/64: LDPC BLER 0.02    RS(255,192) BLER 7.3e-3   N_T = 1e6
                 PER 11/64:           1.3e-4                   9.3e-4         1e6
                 PER 10/64:           0                        6.3e-5         1e7
                 PER  9/64:           0                        2e-6           1e8
    (2000,1000)  PER 24/64:           2.2e-5  RS(250,125)      2.1e-5         2e6
                 PER 23/64:           0                        2e-6           1e7
                 PER 22/64:           0                        0              2e8
(all seven rows, at the paper's own N_T: 3.2e8 frames in about half a minute)

The FPGA's seed is the wall clock (main.cpp:561), so its exact stream is not reproducible: the comparison is statistical,
with BOTH sides' sampling error (exact conditional two-sample test, tests/stat_helpers.py) and the paper's rounding.  The
RS-equivalent BLER is additionally checked against its closed form P[Bin(n_RS, p) > n_RS - k_RS] (tex:217: an MDS block
fails iff more than n-k symbols are erased).  The run is streamed inside the library (O(chunk) memory), N_T as in the
paper.  Both decoder bodies the reference holds are run on the first row; DESIGN.md section 8 records which one
reproduces the table.
"""
import os
import time

import pytest
from scipy.stats import binom

from ldpc_erasure_codes_amd import api
from stat_helpers import consistent_with_rate, consistent_with_reported

pytestmark = pytest.mark.gpu

# code_ind, per64, N_T, LDPC BLER interval the paper's rounded figure stands for, RS BLER interval, paper N_T
ROWS = [
    (1, 12, 1000000, (0.015, 0.025), (7.25e-3, 7.35e-3), 1000000),
    (1, 11, 1000000, (1.25e-4, 1.35e-4), (9.25e-4, 9.35e-4), 1000000),
    (1, 10, 10000000, (0.0, 0.5e-6), (6.25e-5, 6.35e-5), 10000000),   # "0" = printf("%f") of less than 0.5e-6
    (1, 9, 100000000, (0.0, 0.5e-6), (1.5e-6, 2.5e-6), 100000000),
    (0, 24, 2000000, (2.15e-5, 2.25e-5), (2.05e-5, 2.15e-5), 2000000),
    (0, 23, 10000000, (0.0, 0.5e-6), (1.5e-6, 2.5e-6), 10000000),
    (0, 22, 200000000, (0.0, 0.5e-6), (0.0, 0.5e-6), 200000000),     # 4e11 symbols: the 32-bit counter wraps 93 times
]


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def run_trio(ctx, code_ind, per64, nframes, num_iter, seed, halves=False):
    p = api.code_params(code_ind)
    t = time.time()
    ctx.data_in(p[0], seed, per64, code_ind, nframes)
    if halves:
        ctx.ldpc_erasure_decoder_perf_tests(num_iter, code_ind)
    else:
        ctx.ldpc_erasure_decoder(num_iter, code_ind)
    ldpc_err, rs_err = ctx.data_out(code_ind, nframes)
    return ldpc_err, rs_err, time.time() - t


@pytest.mark.parametrize("code_ind,per64,nframes,ldpc_iv,rs_iv,paper_n", ROWS)
def test_table1_row(ctx, code_ind, per64, nframes, ldpc_iv, rs_iv, paper_n):
    p = api.code_params(code_ind)
    n, rs_n, rs_k = p[0], p[4], p[5]
    mult = n // rs_n
    num_iter = 50   # the host's default numItr (main.cpp:99); the table's BLER needs the sweeps run to convergence
    ldpc_err, rs_err, dt = run_trio(ctx, code_ind, per64, nframes, num_iter, seed=20221128 + per64)
    print(f"\nTable I row code {code_ind} PER {per64}/64 N_T {nframes}: LDPC BLER {ldpc_err / nframes:.3g} "
          f"({ldpc_err} errors), RS BLER {rs_err / (mult * nframes):.3g} ({rs_err}), {dt:.2f} s = {nframes / dt:.3g} frames/s")
    ok, pv = consistent_with_reported(ldpc_err, nframes, ldpc_iv[0], ldpc_iv[1], paper_n)
    assert ok, f"LDPC BLER {ldpc_err}/{nframes} vs the paper's {ldpc_iv} on {paper_n} frames: p = {pv:.2e}"
    # RS-equivalent BLER: closed form first (no sampling error on the other side) ...
    exact = binom.sf(rs_n - rs_k, rs_n, per64 / 64.0)
    ok, pv = consistent_with_rate(rs_err, mult * nframes, exact)
    assert ok, f"RS BLER {rs_err}/{mult * nframes} vs exact tail {exact:.4g}: p = {pv:.2e}"
    # ... then the paper's own figure.  Row 11/64 of the paper (9.3e-4) is itself 2.5 sigma above the closed form (9.03e-4)
    ok, pv = consistent_with_reported(rs_err, mult * nframes, rs_iv[0], rs_iv[1], mult * paper_n, alpha=1e-3)
    assert ok, f"RS BLER {rs_err}/{mult * nframes} vs the paper's {rs_iv}: p = {pv:.2e}"


def test_table1_which_decoder_body(ctx):
    """Row PER 12/64 (paper: BLER 0.02) with both bodies of the FPGA decoder kernel and with num_iter 10 / 50.  The in-order
    body (ldpc_erasure_decoder.cl) run to convergence reproduces the table; the two-halves body of
    ldpc_erasure_decoder_perf_tests.cl, as written, does not (its stop rule fires early)."""
    n_t = 200000
    res = {}
    for halves in (False, True):
        for num_iter in (10, 50):
            e, _, dt = run_trio(ctx, 1, 12, n_t, num_iter, seed=99, halves=halves)
            res[(halves, num_iter)] = e / n_t
    print("\nBLER at PER 12/64, (2040,1530):", {("halves" if h else "in-order", it): round(v, 5) for (h, it), v in res.items()})
    assert 0.015 <= res[(False, 50)] <= 0.025          # the table's 0.02
    assert res[(False, 10)] > 0.04                      # ten sweeps are not enough at this PER
    assert res[(True, 50)] > 0.08 and res[(True, 10)] > 0.08   # the two-halves body as written is far off the table


def test_streamed_run_equals_one_chunk(ctx):
    """The run is streamed chunk by chunk: counters and per-frame results do not depend on the chunk size (the erasure
    stream continues across chunk borders exactly where the previous chunk stopped)."""
    nframes = 1000
    want = None
    for chunk in ("", "64", "333"):
        ctx.configure("LDPC_AMD_FPGA_CHUNK", chunk or None)
        try:
            ctx.data_in(2040, 5, 12, 1, nframes)
            ctx.ldpc_erasure_decoder(50, 1)
            left, its = ctx.fpga_frame_stats(nframes)
            got = (ctx.data_out(1, nframes), left.tolist(), its.tolist())
            ctx.data_in(2040, 5, 12, 1, nframes)
            ctx.ldpc_erasure_decoder_perf_tests(10, 1)
            left, its = ctx.fpga_frame_stats(nframes)
            got += (ctx.data_out(1, nframes), left.tolist(), its.tolist())
        finally:
            ctx.configure("LDPC_AMD_FPGA_CHUNK", None)
        if want is None:
            want = got
        assert got == want


def test_trio_call_order_is_checked(ctx):
    """data_out / frame_stats without a decoder call since the last data_in are refused (the FPGA's data_out would block
    on ERROR_STAT), as are mismatching code_ind / numFrames."""
    ctx.data_in(2040, 1, 9, 1, 100)
    ctx.ldpc_erasure_decoder(50, 1)
    assert ctx.data_out(1, 100)[1] >= 0
    ctx.data_in(2040, 1, 9, 1, 5000)      # a larger run, not decoded yet
    with pytest.raises(api.LdpcAmdError):
        ctx.data_out(1, 5000)
    with pytest.raises(api.LdpcAmdError):
        ctx.fpga_frame_stats(5000)
    with pytest.raises(api.LdpcAmdError):
        ctx.ldpc_erasure_decoder(50, 0)   # other code than data_in armed
    ctx.ldpc_erasure_decoder(50, 1)
    with pytest.raises(api.LdpcAmdError):
        ctx.data_out(1, 100)              # numFrames of an earlier run
    assert ctx.data_out(1, 5000)[0] >= 0
    ctx.data_in(2040, 1, 9, 1, 0)          # empty run
    ctx.ldpc_erasure_decoder(50, 1)
    assert ctx.data_out(1, 0) == (0, 0)


def test_paper_figure_bler_curves():
    """The paper's BLER figure for the (2040,1530) code (Latex/LDPC_triangular_2040_1530_Perf_vs_RS.png, produced by
    Matlab/LDPCErasureCodes_MessagePassingAlgSim.m:116,134-245 with PER_vec = [0.14 0.16 0.18 0.2 0.22]): binary code, uniform
    erasures `rand <= PER`, My_LDPC_Erasure_Decoder (50 sweeps) next to My_LDPC_HybridML_Erasure_Decoder (10 sweeps + GF(2)
    elimination) and the RS(255,192)-equivalent count.  Values read off the figure (log axis, +-25 %): message passing 1e-6 at
    0.16, 2.7e-3 at 0.18 (where it crosses the RS curve, tex:164), 0.19 at 0.20, 0.87 at 0.22; RS 2.3e-6, 1.3e-4, 2.8e-3, 2.4e-2, 0.12;
    the MP + ML curve is not visible: no error in the 1e6 / 3.7e5 blocks the script ran at 0.14-0.16 / 0.18.  The only
    reference-produced numbers that involve the ML stage (GF(2) sibling)."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import paper_figure
    tctx = api.Context(0)
    tctx.set_stream(torch.cuda.current_stream().cuda_stream)   # the tool mixes torch ops with library calls: one stream
    rows = {}
    for per, nf in ((0.16, 1000000), (0.18, 400000), (0.20, 50000), (0.22, 20000)):
        r = paper_figure.run(tctx, torch, per, nf, seed=2022)
        rows[per] = r
        print(f"\nfigure PER {per:.2f}: MP {r['mp'] / nf:.3g}  MP+ML {r['ml'] / nf:.3g} ({r['skipped']} skipped)  RS {r['rs'] / r['rs_blocks']:.3g}")
        exact = binom.sf(63, 255, per)
        ok, pv = consistent_with_rate(r["rs"], r["rs_blocks"], exact)
        assert ok, f"RS-equivalent BLER at {per}: {r['rs']}/{r['rs_blocks']} vs {exact:.4g} (p = {pv:.2e})"
    mp = {per: rows[per]["mp"] / rows[per]["frames"] for per in rows}
    rs = {per: rows[per]["rs"] / rows[per]["rs_blocks"] for per in rows}
    assert mp[0.16] <= 1e-5                       # figure: 1e-6
    assert 1.7e-3 <= mp[0.18] <= 3.6e-3           # figure: 2.7e-3
    assert 0.15 <= mp[0.20] <= 0.26               # figure: 0.19
    assert 0.80 <= mp[0.22] <= 0.93               # figure: 0.87
    assert mp[0.16] < rs[0.16] and mp[0.20] > rs[0.20] and 0.5 < mp[0.18] / rs[0.18] < 1.5   # the curves cross at 18 % (tex:164)
    assert rows[0.16]["ml"] == 0 and rows[0.18]["ml"] == 0                                   # MP + ML: no error observed
    assert all(rows[per]["ml"] / rows[per]["frames"] < rs[per] for per in rows)              # ... and below RS at all PERs (tex:164)
    tctx.close()


def test_paper_figure_code_c_qualitative():
    """The paper's second BLER figure, Latex/LDPC_triangular_4080_3060_Perf_vs_RS.png (tex:164), for the (4080,3060) code.  The
    reference does not ship that matrix: this runs the same experiment on the matrix SYNTHESISED by tools/hgen.cpp
    with the reference's construction rules, so only the figure's qualitative statements are checkable: message passing alone
    is better than the RS(255,192)-equivalent code at low PER and worse beyond a crossing near 18-20 %; the MP + ML decoder
    is below RS at every PER ("outperforms the RS code at all PER's")."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import paper_figure
    from ldpc_erasure_codes_amd import codes
    if not codes.have_builtin(3):
        pytest.skip("(4080,3060) not built in")
    tctx = api.Context(0)
    tctx.set_stream(torch.cuda.current_stream().cuda_stream)
    rows = {}
    for per, nf in ((0.16, 250000), (0.18, 100000), (0.20, 20000), (0.22, 5000)):
        r = paper_figure.run(tctx, torch, per, nf, seed=4080, code_ind=3, chunk=50000)
        rows[per] = r
        print(f"\nfigure (4080,3060) [synthesised matrix] PER {per:.2f}: MP {r['mp'] / nf:.3g}  MP+ML {r['ml'] / nf:.3g} "
              f"({r['skipped']} skipped)  RS {r['rs'] / r['rs_blocks']:.3g}")
        exact = binom.sf(63, 255, per)
        ok, pv = consistent_with_rate(r["rs"], r["rs_blocks"], exact)
        assert ok, f"RS-equivalent BLER at {per}: {r['rs']}/{r['rs_blocks']} vs {exact:.4g} (p = {pv:.2e})"
    mp = {per: rows[per]["mp"] / rows[per]["frames"] for per in rows}
    rs = {per: rows[per]["rs"] / rows[per]["rs_blocks"] for per in rows}
    assert mp[0.16] < rs[0.16]                                   # below the crossing: message passing beats RS
    assert mp[0.20] > rs[0.20] and mp[0.22] > rs[0.22]           # beyond it: RS beats message passing alone
    assert all(rows[per]["ml"] / rows[per]["frames"] < rs[per] for per in rows)   # MP + ML below RS at all PERs
    assert rows[0.16]["ml"] == 0 and rows[0.18]["ml"] == 0
    tctx.close()
