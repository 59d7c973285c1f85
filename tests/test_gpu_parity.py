"""GPU parity tests: the HIP path (through the C ABI, include/ldpc_erasure_amd.h) against the CPU oracle
on identical seeded inputs.  Bit-exact: out bytes, sweeps (Matlab `iterations`), residual, status --
including rank-deficient frames, where the reference writes the partially reduced rhs back
(Matlab/My_LDPC_HybridML_NonBinary_Erasure_Decoder.m:87-90,127).

Run on the GPU box with `pytest -m gpu`.  Nothing here reads /root/reference.
"""
import os

import numpy as np
import pytest

from ldpc_erasure_codes_amd import api, codes, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def encode_frames(oracle, code, seed, nframes, S=1):
    oc = oracle.OracleCode(code)
    src = synth.source(seed, 0, nframes, code.k, S)
    if S == 1:
        cw = np.stack([oc.encode(src[f, :, 0]) for f in range(nframes)])
        return oc, src[:, :, 0], cw
    cw = np.stack([oc.encode(src[f]) for f in range(nframes)])
    return oc, src, cw


def corrupt(cw, erased, fill=0xA5):
    sym = cw.copy()
    sym[erased.astype(bool)] = fill  # payload of erased symbols must be ignored
    return sym


def check_s1(ctx, oracle, code, handle, cw, erased, max_sweeps=10, do_ml=1):
    oc = oracle.OracleCode(code)
    sym = corrupt(cw, erased)
    out, sw, res, st = ctx.decode(handle, sym, erased, max_sweeps=max_sweeps, do_ml=do_ml)
    o_out, o_sw, o_res, o_st = oc.decode_batch_s1(sym, erased, itenum=max_sweeps, do_ml=do_ml)
    assert np.array_equal(sw, o_sw), "sweeps"
    assert np.array_equal(res, o_res), "residual"
    assert np.array_equal(st, o_st), "status"
    bad = np.nonzero((out != o_out).any(axis=1))[0]
    assert bad.size == 0, f"frames differ: {bad[:10]} status {o_st[bad[:10]]}"
    return o_st, o_sw


def test_selftest_gf_primitives(ctx):
    ctx.selftest()


@pytest.mark.parametrize("per,nframes", [(0.10, 256), (0.1406, 128), (0.18, 128), (0.20, 96), (0.22, 64), (0.235, 48)])
def test_code_a_s1_uniform(ctx, oracle, code_a, per, nframes):
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    _, _, cw = encode_frames(oracle, code_a, 11, nframes)
    erased = synth.erasures_uniform(1000 + int(per * 1e4), 0, nframes, code_a.n, per)
    st, sw = check_s1(ctx, oracle, code_a, h, cw, erased)
    if per <= 0.15:
        assert (st == 0).all()
    if per >= 0.22:
        assert (st == api.ST_ML_SOLVED).any() or (st == api.ST_ML_RANKDEF).any()


def test_code_a_s1_covers_every_status(ctx, oracle, code_a):
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    nframes = 160
    _, _, cw = encode_frames(oracle, code_a, 12, nframes)
    erased = np.concatenate([synth.erasures_uniform(5, 0, 40, code_a.n, p) for p in (0.12, 0.21, 0.235, 0.27)])
    st, sw = check_s1(ctx, oracle, code_a, h, cw, erased)
    assert set(np.unique(st)) == {0, 1, 2, 3}, np.unique(st)


def test_code_a_s1_bursty_channel(ctx, oracle, code_a):
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    nframes = 128
    _, _, cw = encode_frames(oracle, code_a, 13, nframes)
    for alpha, beta in ((0.10, 0.4), (0.12, 0.8)):
        erased = synth.erasures_bursty(21, 0, nframes, code_a.n, alpha, beta, 10.0)
        check_s1(ctx, oracle, code_a, h, cw, erased)


@pytest.mark.parametrize("max_sweeps,do_ml", [(1, 1), (2, 1), (3, 0), (10, 0), (50, 0), (50, 1)])
def test_code_a_s1_sweep_cap_and_ml_switch(ctx, oracle, code_a, max_sweeps, do_ml):
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    nframes = 48
    _, _, cw = encode_frames(oracle, code_a, 14, nframes)
    erased = np.concatenate([synth.erasures_uniform(6, 0, 24, code_a.n, p) for p in (0.12, 0.2)])
    check_s1(ctx, oracle, code_a, h, cw, erased, max_sweeps=max_sweeps, do_ml=do_ml)


def test_code_a_s1_edge_patterns(ctx, oracle, code_a):
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    _, _, cw = encode_frames(oracle, code_a, 15, 8)
    n, k, m = code_a.n, code_a.k, code_a.m
    erased = np.zeros((8, n), dtype=np.uint8)
    erased[1, 17] = 1                       # single erasure
    erased[2, k:] = 1                       # every parity symbol: one in-order sweep re-encodes
    erased[3, :600] = 1                     # more erasures than checks: ML cannot run
    erased[4, :m - 1] = 1                   # n-k-1 erasures, contiguous
    erased[5, n - 1] = 1                    # last parity symbol only
    erased[6, ::7] = 1
    erased[7, 100:100 + m] = 1              # exactly n-k erasures
    st, sw = check_s1(ctx, oracle, code_a, h, cw, erased)
    assert sw[0] == 1 and st[0] == 0        # no erasure: one sweep still runs
    assert sw[2] == 1 and st[2] == 0
    assert st[3] == api.ST_ML_SKIPPED


@pytest.mark.parametrize("code_ind,pers", [(0, (0.3, 0.4, 0.45)), (2, (0.3, 0.4, 0.45))])
def test_rate_half_codes_s1(ctx, oracle, code_ind, pers):
    code = codes.load_builtin(code_ind)
    h = ctx.load_builtin_code(code_ind, codes.DEFAULT_COEF_SEED[code_ind])
    rp, cols, coefs = ctx.code_csr(h)
    assert np.array_equal(coefs, code.coefs) and np.array_equal(cols, code.cols)
    nframes = 24
    _, _, cw = encode_frames(oracle, code, 16, nframes)
    erased = np.concatenate([synth.erasures_uniform(7, 0, 8, code.n, p) for p in pers])
    check_s1(ctx, oracle, code, h, cw, erased)


@pytest.mark.parametrize("n,k,rowdeg,per", [(60, 36, 5, 0.3), (200, 120, 7, 0.3), (130, 66, 20, 0.25), (64, 1, 2, 0.4),
                                            (700, 630, 24, 0.07)])
def test_custom_codes_s1(ctx, oracle, n, k, rowdeg, per):
    """register_code with small hand-made triangle codes: exercises the degree buckets 8/16/24, m not a
    multiple of 64, n not a multiple of 4."""
    rng = np.random.default_rng(n * 1000 + k)
    m = n - k
    H = np.zeros((m, n), dtype=np.uint8)
    for i in range(m):
        c = rng.choice(k + i, size=min(rowdeg - 1, k + i), replace=False)
        H[i, c] = rng.integers(1, 256, size=c.size)
        H[i, k + i] = rng.integers(1, 256)
    code = codes.from_dense(H, k)
    h = ctx.register_code(code)
    nframes = 64
    _, _, cw = encode_frames(oracle, code, 17, nframes)
    erased = synth.erasures_uniform(8, 0, nframes, n, per)
    st, _ = check_s1(ctx, oracle, code, h, cw, erased)
    # encoder parity too
    src = synth.source(17, 0, nframes, k, 1)[:, :, 0]
    assert np.array_equal(ctx.encode(h, src), cw)


@pytest.mark.parametrize("n,k,rowdeg,S,per", [(60, 36, 5, 16, 0.35), (200, 120, 7, 48, 0.36), (130, 66, 20, 272, 0.3),
                                                (333, 250, 9, 32, 0.22), (700, 630, 24, 64, 0.085), (96, 32, 4, 1040, 0.5)])
def test_custom_codes_packets_incl_rank_deficient(ctx, oracle, n, k, rowdeg, S, per):
    """Packet mode on hand-made codes (degree buckets 8/14/16/24, m and n of no convenient multiple, S that is not a
    multiple of 256, so every row-piece width of the packet kernel is used), erasure rates that push frames into the
    ML stage and some of them into rank deficiency; the input is NOT a codeword in half of the frames (random packets),
    so the junk the reference leaves behind depends on every byte.  Every byte lane must equal the oracle's."""
    rng = np.random.default_rng(n * 7 + S)
    m = n - k
    H = np.zeros((m, n), dtype=np.uint8)
    for i in range(m):
        c = rng.choice(k + i, size=min(rowdeg - 1, k + i), replace=False)
        H[i, c] = rng.integers(1, 256, size=c.size)
        H[i, k + i] = rng.integers(1, 256)
    code = codes.from_dense(H, k)
    h = ctx.register_code(code)
    oc = oracle.OracleCode(code)
    nframes = 12
    src = synth.source(23, 0, nframes, k, S)
    cw = ctx.encode(h, src)
    assert np.array_equal(cw[1], oc.encode(src[1]))
    sym = cw.copy()
    sym[nframes // 2:] = rng.integers(0, 256, size=sym[nframes // 2:].shape, dtype=np.uint8)   # not codewords
    erased = synth.erasures_uniform(24, 0, nframes, n, per)
    erased[0] = 0                                                    # nothing erased
    erased[1] = 0; erased[1, k:] = 1                                 # all parity erased
    sym = corrupt(sym, erased, fill=0x3C)
    out, sw, res, st = ctx.decode(h, sym, erased)
    seen = set()
    for f in range(nframes):
        o_out, o_er, o_it, info, rc = oc.decode_packets(sym[f], erased[f])
        want_st = 0 if info[0] == 0 else (3 if (rc == -2 or not info[1]) else (2 if info[2] else 1))
        seen.add(want_st)
        assert sw[f] == o_it and res[f] == info[0] and st[f] == want_st, (f, sw[f], o_it, res[f], info, st[f], want_st)
        if want_st != 3:   # (skipped frames: the reference harness does not call the decoder, out is unspecified there)
            assert np.array_equal(out[f], o_out), f"frame {f} (status {want_st})"
    assert 1 in seen or 2 in seen   # the ML stage was reached


@pytest.mark.parametrize("S", [16, 64, 1024, 2048])
def test_code_a_packets(ctx, oracle, code_a, S):
    """Packet mode: every byte lane equals the Matlab-exact scalar decode of that lane."""
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    nframes = 6 if S >= 1024 else 10
    oc = oracle.OracleCode(code_a)
    # build codewords on the GPU encoder and cross-check one against the oracle encoder
    src = synth.source(18, 0, nframes, code_a.k, S)
    cw = ctx.encode(h, src)
    assert np.array_equal(cw[0], oc.encode(src[0]))
    pers = [0.1, 0.1406, 0.18, 0.21, 0.235, 0.27, 0.0, 0.2, 0.22, 0.19][:nframes]
    erased = np.concatenate([synth.erasures_uniform(9 + i, i, 1, code_a.n, p) for i, p in enumerate(pers)])
    sym = corrupt(cw, erased)
    out, sw, res, st = ctx.decode(h, sym, erased)
    for f in range(nframes):
        o_out, o_er, o_it, info, rc = oc.decode_packets(sym[f], erased[f])
        assert sw[f] == o_it and res[f] == info[0], (f, sw[f], o_it, res[f], info)
        want_st = 0 if info[0] == 0 else (3 if (rc == -2 or not info[1]) else (2 if info[2] else 1))
        assert st[f] == want_st
        assert np.array_equal(out[f], o_out), f"frame {f} (status {want_st})"
        if want_st in (0, 1):
            assert np.array_equal(out[f], cw[f])


def test_code_b_packets_and_encoder(ctx, oracle, code_b):
    h = ctx.load_builtin_code(2, codes.DEFAULT_COEF_SEED[2])
    oc = oracle.OracleCode(code_b)
    S, nframes = 32, 4
    src = synth.source(19, 0, nframes, code_b.k, S)
    cw = ctx.encode(h, src)
    for f in range(nframes):
        assert np.array_equal(cw[f], oc.encode(src[f]))
    erased = np.concatenate([synth.erasures_uniform(30 + i, i, 1, code_b.n, p) for i, p in enumerate((0.3, 0.42, 0.45, 0.47))])
    sym = corrupt(cw, erased)
    out, sw, res, st = ctx.decode(h, sym, erased)
    for f in range(nframes):
        o_out, o_er, o_it, info, rc = oc.decode_packets(sym[f], erased[f])
        assert sw[f] == o_it and res[f] == info[0]
        assert np.array_equal(out[f], o_out), f


def test_binary_code_matches_binary_reference_decoder(ctx, oracle, code_a):
    """cfg 1 semantics: My_LDPC_Erasure_Decoder (itenum = 50, XOR) == coefficient-1 code, do_ml = 0."""
    cb = code_a.binary()
    h = ctx.load_builtin_code(1, 0)
    oc = oracle.OracleCode(cb)
    nframes = 32
    rng = np.random.default_rng(3)
    src = rng.integers(0, 2, size=(nframes, cb.k)).astype(np.uint8)
    cw = np.stack([oc.encode(s) for s in src])
    erased = synth.erasures_uniform(40, 0, nframes, cb.n, 9 / 64)
    erased[16:] = synth.erasures_uniform(41, 0, 16, cb.n, 0.2)
    sym = corrupt(cw, erased, fill=1)
    out, sw, res, st = ctx.decode(h, sym, erased, max_sweeps=50, do_ml=0)
    for f in range(nframes):
        recv = cw[f].astype(np.int16)
        recv[erased[f].astype(bool)] = -1
        msg, it = oc.binary_mp(recv, itenum=50)
        assert sw[f] == it
        assert res[f] == (msg == -1).sum()
        want = msg.copy()
        want[want == -1] = 0
        assert np.array_equal(out[f], want.astype(np.uint8))


def test_device_pointer_mode_matches_host_mode(ctx, oracle, code_a):
    torch = pytest.importorskip("torch")
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    nframes, S = 8, 64
    src = synth.source(20, 0, nframes, code_a.k, S)
    cw = ctx.encode(h, src)
    erased = synth.erasures_uniform(50, 0, nframes, code_a.n, 0.15)
    sym = corrupt(cw, erased)
    out_h, sw_h, res_h, st_h = ctx.decode(h, sym, erased)
    d_sym = torch.from_numpy(sym).cuda()
    d_er = torch.from_numpy(erased).cuda()
    torch.cuda.synchronize()  # the context has its own stream: order torch's uploads before the decode
    out_d, sw_d, res_d, st_d = ctx.decode(h, d_sym, d_er)
    ctx.synchronize()
    assert np.array_equal(out_d.cpu().numpy(), out_h)
    assert np.array_equal(sw_d.cpu().numpy(), sw_h)
    assert np.array_equal(st_d.cpu().numpy(), st_h)
    # device generators == host generators
    d_src = torch.empty((nframes, code_a.k, S), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.synth_source(20, 0, nframes, code_a.k, S, d_src)
    d_e = torch.empty((nframes, code_a.n), dtype=torch.uint8, device="cuda")
    ctx.synth_erasures_uniform(50, 0, nframes, code_a.n, 0.15, d_e)
    ctx.synchronize()
    assert np.array_equal(d_src.cpu().numpy(), src)
    assert np.array_equal(d_e.cpu().numpy(), erased)


def test_full_batch_round_trip_cfg2(code_a):
    """BASELINE cfg 2 at full size (4096 frames, uniform 10 %): encode -> erase -> decode == codeword,
    S = 1 and S = 64; sweeps histogram as measured in SURVEY.md 7.3 (2 or 3 sweeps, no ML)."""
    torch = pytest.importorskip("torch")
    ctx = api.Context(0)
    # device-pointer calls are asynchronous on the context's stream: share torch's stream so that the
    # tensor ops below and the library's kernels are ordered
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    nframes = 4096
    for S in (1, 64):
        src = torch.empty((nframes, code_a.k, S), dtype=torch.uint8, device="cuda")
        ctx.synth_source(77, 0, nframes, code_a.k, S, src)
        if S == 1:
            src = src.reshape(nframes, code_a.k)
        cw = ctx.encode(h, src)
        er = torch.empty((nframes, code_a.n), dtype=torch.uint8, device="cuda")
        ctx.synth_erasures_uniform(78, 0, nframes, code_a.n, 0.10, er)
        sym = cw.clone()
        sym[er.bool()] = 0x5A
        out, sw, res, st = ctx.decode(h, sym, er)
        ctx.synchronize()
        assert torch.equal(out, cw)
        assert int(st.max()) == 0 and int(res.max()) == 0
        hist = torch.bincount(sw, minlength=5).cpu().numpy()
        assert hist[2] + hist[3] + hist[4] == nframes and hist[2] > hist[3] > hist[4]
    ctx.close()


# ------------------------------------------------------------------------------------ Reed-Solomon
@pytest.mark.parametrize("n,k", [(7, 5), (255, 223), (255, 192), (250, 125)])
def test_rs_decode_s1(ctx, oracle, n, k):
    rs = ctx.rs_create(n, k)
    G = ctx.rs_generator(rs, n, k)
    assert np.array_equal(G, oracle.rs_generator(n, k))
    rng = np.random.default_rng(n * k)
    B = 64
    src = rng.integers(0, 256, size=(B, k)).astype(np.uint8)
    cw = ctx.rs_encode(rs, n, k, src)
    idx = np.zeros((B, k), dtype=np.uint16)
    for b in range(B):
        assert np.array_equal(cw[b], oracle.rs_encode(G, src[b]))
        nerase = [0, 1, n - k, (n - k) // 2][b % 4] if b < 8 else int(rng.integers(0, n - k + 1))
        idx[b] = np.sort(rng.permutation(n)[: n - nerase])[:k]
    val = np.take_along_axis(cw, idx.astype(np.int64), axis=1)
    msg = ctx.rs_decode(rs, idx, val)
    for b in range(B):
        o, rc = oracle.rs_decode(G, idx[b], val[b])
        assert rc == 0 and np.array_equal(o, src[b])
        assert np.array_equal(msg[b], o), b


@pytest.mark.parametrize("S", [16, 1024])
def test_rs_decode_packets(ctx, oracle, S):
    n, k = 255, 223
    rs = ctx.rs_create(n, k)
    G = ctx.rs_generator(rs, n, k)
    rng = np.random.default_rng(S)
    B = 6
    src = rng.integers(0, 256, size=(B, k, S)).astype(np.uint8)
    cw = ctx.rs_encode(rs, n, k, src)
    for lane in (0, S - 1):
        assert np.array_equal(cw[0, :, lane], oracle.rs_encode(G, np.ascontiguousarray(src[0, :, lane])))
    idx = np.zeros((B, k), dtype=np.uint16)
    for b in range(B):
        idx[b] = np.sort(rng.permutation(n)[: n - [0, 5, 32, 17, 31, 1][b]])[:k]
    val = np.stack([cw[b, idx[b].astype(np.int64)] for b in range(B)])
    msg = ctx.rs_decode(rs, idx, val)
    assert np.array_equal(msg, src)
    for b in (2, 3):
        for lane in (0, S // 2):
            o, rc = oracle.rs_decode(G, idx[b], np.ascontiguousarray(val[b, :, lane]))
            assert np.array_equal(msg[b, :, lane], o)


# ------------------------------------------------------------------------------------ FPGA harness trio
@pytest.mark.parametrize("code_ind,per64", [(1, 9), (1, 12), (0, 23)])
def test_fpga_harness_statistics(ctx, oracle, code_ind, per64):
    """data_in -> ldpc_erasure_decoder -> data_out with the reference's argument lists
    (OpenCL/host/src/main.cpp:578-604); counters as in ldpc_erasure_decoder_perf_tests.cl:70-80,215-236."""
    code = codes.load_builtin(code_ind, binary=True)
    p = api.code_params(code_ind)
    n, k, rs_n, rs_k = p[0], p[1], p[4], p[5]
    nframes, seed, num_iter = 200, 4242, 50
    ctx.data_in(n, seed, per64, code_ind, nframes)
    ctx.ldpc_erasure_decoder(num_iter, code_ind)
    ldpc_err, rs_err = ctx.data_out(code_ind, nframes)
    erased = oracle.fpga_data_in_erasures(seed, per64, nframes, n)  # threefry stream of the FPGA kernel
    assert np.array_equal(erased, synth.fpga_erasures(seed, per64, nframes, n))
    oc = oracle.OracleCode(code)
    want_ldpc = 0
    for f in range(nframes):
        recv = np.zeros(n, dtype=np.int16)
        recv[erased[f].astype(bool)] = -1
        msg, it = oc.binary_mp(recv, itenum=num_iter)
        want_ldpc += int((msg[:k] == -1).any())
    blocks = erased[:, : (n // rs_n) * rs_n].reshape(nframes, n // rs_n, rs_n).sum(axis=2)
    want_rs = int((blocks > rs_n - rs_k).sum())
    assert (ldpc_err, rs_err) == (want_ldpc, want_rs)


@pytest.mark.parametrize("code_ind,per64,num_iter", [(1, 9, 50), (1, 12, 10), (0, 23, 50), (1, 9, 2)])
def test_fpga_perf_tests_decoder_two_halves(ctx, oracle, code_ind, per64, num_iter):
    """The other body of the FPGA decoder kernel, OpenCL/device/ldpc_erasure_decoder_perf_tests.cl:56-236 (two copies,
    half the checks each, merge, stop when num_current_correct == k): per-frame systematic erasures left and
    iteration counts against the oracle's restatement, premature stops included, and the ERROR_STAT totals."""
    code = codes.load_builtin(code_ind, binary=True)
    p = api.code_params(code_ind)
    n = p[0]
    nframes, seed = 300, 777
    ctx.data_in(n, seed, per64, code_ind, nframes)
    ctx.ldpc_erasure_decoder_perf_tests(num_iter, code_ind)
    left, its = ctx.fpga_frame_stats(nframes)
    ldpc_err, _ = ctx.data_out(code_ind, nframes)
    erased = synth.fpga_erasures(seed, per64, nframes, n)
    oc = oracle.OracleCode(code)
    want = [oc.fpga_perf_decoder(erased[f], num_iter)[:2] for f in range(nframes)]
    assert np.array_equal(left, np.array([w[0] for w in want]))
    assert np.array_equal(its, np.array([w[1] for w in want]))
    assert ldpc_err == sum(w[0] > 0 for w in want)
    # the in-order decoder of ldpc_erasure_decoder.cl still reports through the same per-frame interface
    ctx.ldpc_erasure_decoder(num_iter, code_ind)
    left2, its2 = ctx.fpga_frame_stats(nframes)
    for f in range(0, nframes, 37):
        recv = np.zeros(n, dtype=np.int16)
        recv[erased[f].astype(bool)] = -1
        msg, it = oc.binary_mp(recv, itenum=num_iter)
        assert left2[f] == int((msg[: p[1]] == -1).sum()) and its2[f] == it
