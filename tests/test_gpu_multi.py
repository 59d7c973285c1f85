"""The C-level multi-device layer on the GPU (include/ldpc_erasure_amd_multi.h, csrc/multi.hip): N ranks -- one context and one
host thread each -- on the devices of the box (on a one-GPU box all ranks share device 0 and the final gather degenerates to
device-to-device copies).  Every result must equal the single-context result: frames are independent
(Matlab/ErasureCodes_NonBinaryLDPCSim.m:218), so a sharded run is the same run."""
import os
import subprocess

import numpy as np
import pytest

from ldpc_erasure_codes_amd import api, codes, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _batch(code, F, S, seed, lo=0.05, hi=0.26):
    src = synth.source(seed, 0, F, code.k, S)
    pers = np.linspace(lo, hi, F)
    era = np.concatenate([synth.erasures_uniform(seed + 1 + i, i, 1, code.n, float(p)) for i, p in enumerate(pers)])
    return (src if S > 1 else src[:, :, 0]), era


@pytest.mark.parametrize("nranks,F,S", [(3, 50, 64), (2, 33, 1), (4, 3, 256), (3, 1, 1)])
def test_group_decode_host_pointers_equals_one_context(code_a, oracle, nranks, F, S):
    src, era = _batch(code_a, F, S, 1200 + F)
    with api.Context(0) as c:
        h = c.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
        cw = c.encode(h, src)
        sym = cw.copy()
        sym[era.astype(bool)] = 0x5A
        ref = c.decode(h, sym, era)
    with api.Group(nranks) as g:       # ragged shards, and (F < nranks) empty ones
        gh = g.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
        got = g.decode(gh, sym, era)
    for a_, b_, what in zip(got, ref, ("out", "sweeps", "residual", "status")):
        assert np.array_equal(a_, b_), (nranks, F, S, what)
    oc = oracle.OracleCode(code_a)     # and the single-context result is the oracle's (first and last frame)
    for f in {0, F - 1}:
        if S == 1:
            o = oc.decode_batch_s1(sym[f:f + 1], era[f:f + 1])
            assert np.array_equal(ref[0][f], o[0][0]) and ref[1][f] == o[1][0] and ref[3][f] == o[3][0]
        else:
            o, _, it, info, rc = oc.decode_packets(sym[f], era[f])
            assert np.array_equal(ref[0][f], o) and ref[1][f] == it


def test_group_decode_resident_gathers_to_rank0_device(code_a):
    import torch
    nranks, F, S = 3, 40, 128
    src, era_np = _batch(code_a, F, S, 77)
    dev = torch.device("cuda", 0)
    with api.Context(0) as c:
        h = c.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
        cw = c.encode(h, src)
        sym_np = cw.copy()
        sym_np[era_np.astype(bool)] = 0xC3
        ref = c.decode(h, sym_np, era_np)
    with api.Group(nranks) as g:
        gh = g.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
        sym, era, out, words = [], [], [], []
        for r in range(nranks):
            f0, cnt = api.shard_frames(F, nranks, r)
            assert g.device(r) == r % torch.cuda.device_count()
            d = torch.device("cuda", g.device(r))
            sym.append(torch.from_numpy(sym_np[f0:f0 + cnt]).to(d).contiguous())
            era.append(torch.from_numpy(era_np[f0:f0 + cnt]).to(d).contiguous())
            out.append(torch.empty_like(sym[-1]))
            words.append(torch.empty((3, cnt), dtype=torch.int32, device=d))
        gw = torch.full((3, F), -7, dtype=torch.int32, device=dev)
        gout = torch.zeros((F, code_a.n, S), dtype=torch.uint8, device=dev)
        dms, gms = g.decode_resident(gh, S, F, sym, era, out, words, gathered_words=gw, gathered_out=gout)
        assert dms > 0 and gms > 0
        gw_np = gw.cpu().numpy()
        assert np.array_equal(gw_np[0], ref[1]) and np.array_equal(gw_np[1], ref[2]) and np.array_equal(gw_np[2], ref[3])
        assert np.array_equal(gout.cpu().numpy(), ref[0])
        dms, gms = g.decode_resident(gh, S, F, sym, era, out, words)        # no gather: outputs stay sharded
        assert np.array_equal(torch.cat(out).cpu().numpy(), ref[0])


@pytest.mark.parametrize("perf_body", [False, True])
def test_sharded_fpga_run_equals_the_single_device_run(perf_body):
    """data_in / ldpc_erasure_decoder / data_out over 3 ranks: every rank draws its block of the SAME threefry stream, so the summed
    counters are the single-device run's (both decoder bodies; a frame count 3 does not divide)."""
    seed, per64, code_ind, nf, iters = 4242, 12, 1, 20000 + 1, 50
    with api.Context(0) as c:
        c.data_in(2040, seed, per64, code_ind, nf)
        (c.ldpc_erasure_decoder_perf_tests if perf_body else c.ldpc_erasure_decoder)(iters, code_ind)
        one = c.data_out(code_ind, nf)
    with api.Group(3) as g:
        many = g.fpga_run(2040, seed, per64, code_ind, nf, iters, perf_tests_body=perf_body)
    assert tuple(one) == tuple(many), (one, many)
    assert many[0] > 0 and many[1] > 0


def test_c_bench_and_host_harness_with_ranks():
    with api.Group(2) as g:
        r = g.bench_resident(1, codes.DEFAULT_COEF_SEED[1], 1024, 96, steps=3)
        assert r["verified"] == 1.0 and r["frames_per_s"] > 0 and r["gather_ms"] > 0, r
    exe = os.path.join(ROOT, "ldpc_erasure_codes_amd", "host", "ldpc_erasure_decoder_host")
    if not os.path.exists(exe):
        pytest.skip("host harness not built")
    r = subprocess.run([exe, "-h", "-c", "1", "-p", "12", "-n", "300001", "-i", "50", "-g", "3", "-b", "64"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "PASSED" in r.stdout and "3 ranks on devices" in r.stdout
    line = [ln for ln in r.stdout.splitlines() if "frame error rate" in ln][0]
    fer = float(line.split("frame error rate is:")[1].split(",")[0])
    assert 0.015 < fer < 0.025, line                       # Table I, PER 12/64 (tex:207): 0.02
    shard_lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("rank ")]
    assert len(shard_lines) == 3 and "[0, 100001)" in shard_lines[0] and "[200001, 300001)" in shard_lines[2]
    pay = [ln for ln in r.stdout.splitlines() if ln.startswith("Payload run:")][0]
    assert "every frame equals its codeword" in pay and "3 rank(s) x 64 frames" in pay
