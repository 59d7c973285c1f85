"""Pins the CPU oracle (oracle/oracle.c) against everything the reference holds for the hot path
(SURVEY.md section 8c): the complete GF(256) tables, the three H matrices / the OpenCL code ROM, the
paper's (6,3) worked example -- and against a second, independent 1-based transliteration of the same
Matlab files (tests/matlab_literal.py).  CPU only.
"""
import os

import numpy as np
import pytest

import matlab_literal as ml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from ldpc_erasure_codes_amd import codes, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ---------------------------------------------------------------- a6: GF tables (complete KAT)
def test_gf_tables_equal_reference_mat(oracle):
    ref = np.load(os.path.join(GOLD, "gf256_tables_ref.npz"))
    t = oracle.gf_tables()
    assert np.array_equal(t["add"], ref["GF_add_lookup"])
    assert np.array_equal(t["mult"], ref["GF_mult_lookup"])
    assert np.array_equal(t["inv"], ref["GF_inv_lookup"])


def test_gf_tables_poly_is_0x171_not_0x11d(oracle):
    ref = np.load(os.path.join(GOLD, "gf256_tables_ref.npz"))
    assert not np.array_equal(oracle.gf_tables(0x11D)["mult"], ref["GF_mult_lookup"])
    t = oracle.gf_tables(0x171)
    assert np.array_equal(t["mult"], ref["GF_mult_lookup"])
    # spot values recorded in SURVEY.md appendix B
    assert t["inv"][:8].tolist() == [1, 184, 208, 92, 159, 104, 134, 46]
    assert t["mult"][2, 2] == 4 and t["mult"][2, 128] == 113
    x = np.arange(1, 256)
    assert np.all(t["mult"][x, t["inv"][x - 1]] == 1)


# ---------------------------------------------------------------- a7: code tables
@pytest.mark.parametrize("code_ind,first,last", [(0, 0, 999), (1, 1000, 1509)])
def test_code_fixture_equals_opencl_code_rom(code_ind, first, last):
    rom = np.load(os.path.join(GOLD, "code_rom_ref.npz"))
    params, vl = rom["ldpc_params"], rom["vlist_master"]
    c = codes.load_builtin(code_ind)
    assert params[code_ind, 0] == c.n and params[code_ind, 1] == c.k
    assert (params[code_ind, 2], params[code_ind, 3]) == (first, last)
    assert (params[code_ind, 4], params[code_ind, 5]) == (c.rs_n, c.rs_k)
    for r in range(c.m):
        row = vl[first + r]
        deg = int(row[0])
        s, e = int(c.row_ptr[r]), int(c.row_ptr[r + 1])
        assert deg == e - s
        assert np.array_equal(row[1:1 + deg].astype(int) - 1, c.cols[s:e].astype(int))  # ROM is 1-based
        assert np.all(row[1 + deg:] == 0)


def test_code_a_known_rows():
    # first / last Vlist row of code A (1-based)
    c = codes.load_builtin(1)
    first = (c.cols[c.row_ptr[0]:c.row_ptr[1]].astype(int) + 1).tolist()
    assert first == [444, 502, 517, 679, 700, 722, 790, 850, 890, 1001, 1032, 1251, 1531]
    last = (c.cols[c.row_ptr[-2]:c.row_ptr[-1]].astype(int) + 1).tolist()
    assert last == [2039, 2040]


@pytest.mark.parametrize("code_ind,shape", [(0, (1000, 2000, 5998)), (1, (510, 2040, 6628)), (2, (2000, 4000, 12002))])
def test_code_structure(code_ind, shape):
    c = codes.load_builtin(code_ind)
    assert (c.m, c.n, c.nnz) == shape
    # triangle form: the last non-zero of row i is column k+i; coefficients are 1..255
    assert np.array_equal(c.cols[c.row_ptr[1:] - 1], c.k + np.arange(c.m))
    assert c.coefs.min() >= 1


# ---------------------------------------------------------------- synthetic input generator
def test_synth_matches_c_header(oracle):
    assert np.array_equal(synth.coefs(2040, 6628), oracle.synth_coefs(2040, 6628))
    assert np.array_equal(synth.source(7, 3, 2, 50, 4), oracle.synth_source(7, 3, 2, 50, 4))
    for per in (0.0, 0.1, 0.1406, 0.5, 1.0):
        assert np.array_equal(synth.erasures_uniform(9, 5, 3, 200, per), oracle.synth_erasures_uniform(9, 5, 3, 200, per))
    a = synth.erasures_bursty(11, 2, 3, 300, 0.1, 0.4, 10)
    b = oracle.synth_erasures_bursty(11, 2, 3, 300, 0.1, 0.4, 10)
    assert np.array_equal(a, b)
    c = synth.coefs(1, 100000)
    assert c.min() == 1 and c.max() == 255


def test_bursty_step_matches_literal(oracle):
    import ctypes as C
    rng = np.random.default_rng(0)
    for _ in range(2000):
        st = int(rng.integers(0, 2))
        a, b = float(rng.random() * 0.3), float(rng.random())
        r1, r2 = float(rng.random()), float(rng.random())
        ns = C.c_int(0)
        e = oracle.lib().oracle_bursty_channel_step(st, a, b, 10.0, r1, r2, C.byref(ns))
        assert (e, ns.value) == ml.bursty_step(st, a, b, 10.0, r1, r2)


def test_bursty_mean_per_formula(oracle):
    # Matlab/ErasureCodes_NonBinaryLDPCSim.m:137 PER = a/(1+1/bias) + (1 - 1/(1+1/bias)) b
    alpha, beta, bias = 0.1, 0.4, 10.0
    era = oracle.synth_erasures_bursty(3, 0, 200, 2040, alpha, beta, bias)
    per = (1 / (1 + 1 / bias)) * alpha + (1 - 1 / (1 + 1 / bias)) * beta
    assert abs(era.mean() - per) < 0.01


# ---------------------------------------------------------------- paper's worked example
def test_paper_6_3_example(oracle):
    # Latex/Milcom_2022_ErasureCodes.tex:83-102: p1 = s2, p2 = s1 ^ s3, p3 = s3 ^ p1
    H = np.array([[0, 1, 0, 1, 0, 0], [1, 0, 1, 0, 1, 0], [0, 0, 1, 1, 0, 1]], dtype=np.uint8)
    c = codes.from_dense(H, 3)
    oc = oracle.OracleCode(c)
    for s in range(8):
        s1, s2, s3 = (s >> 2) & 1, (s >> 1) & 1, s & 1
        cw = oc.encode(np.array([s1, s2, s3], dtype=np.uint8))
        assert cw.tolist() == [s1, s2, s3, s2, s1 ^ s3, s3 ^ s2]
        assert np.array_equal(ml.encode(H.astype(int), [s1, s2, s3]), cw)


# ---------------------------------------------------------------- oracle vs second restatement
def small_code(seed, n=60, k=36, rowdeg=5):
    """Random small triangle-form GF(256) code: row i has rowdeg-1 random earlier columns + diagonal k+i."""
    rng = np.random.default_rng(seed)
    m = n - k
    H = np.zeros((m, n), dtype=np.uint8)
    for i in range(m):
        cols = rng.choice(k + i, size=min(rowdeg - 1, k + i), replace=False)
        H[i, cols] = rng.integers(1, 256, size=cols.size)
        H[i, k + i] = rng.integers(1, 256)
    return H


@pytest.mark.parametrize("seed", range(6))
def test_encoder_matches_literal_and_satisfies_checks(oracle, seed):
    H = small_code(seed)
    c = codes.from_dense(H, 36)
    oc = oracle.OracleCode(c)
    rng = np.random.default_rng(100 + seed)
    src = rng.integers(0, 256, size=36).astype(np.uint8)
    cw = oc.encode(src)
    assert np.array_equal(cw, ml.encode(H.astype(int), src))
    # H c^T = 0 over GF(256)
    t = oracle.gf_tables()
    for r in range(H.shape[0]):
        s = 0
        for j in np.nonzero(H[r])[0]:
            s ^= int(t["mult"][H[r, j], cw[j]])
        assert s == 0


@pytest.mark.parametrize("seed,per", [(0, 0.1), (1, 0.2), (2, 0.3), (3, 0.35), (4, 0.4), (5, 0.45), (6, 0.3), (7, 0.38)])
def test_decoder_matches_literal_small_codes(oracle, seed, per):
    """All three outputs (Msg incl. junk on rank-deficient frames, iterations, dont_do_jordan) agree."""
    H = small_code(seed)
    c = codes.from_dense(H, 36)
    oc = oracle.OracleCode(c)
    rng = np.random.default_rng(200 + seed)
    seen = set()
    for trial in range(12):
        src = rng.integers(0, 256, size=36).astype(np.uint8)
        cw = oc.encode(src).astype(np.int16)
        recv = cw.copy()
        recv[rng.random(60) <= per] = -1
        if (recv == -1).sum() > H.shape[0]:
            continue
        for itenum in (10, 2):
            msg, it, info, rc = oc.decode(recv, itenum=itenum)
            msg2, it2, dj2 = ml.hybridml_nonbinary_decode(recv.astype(int), H.astype(int), itenum=itenum)
            assert rc == 0
            assert np.array_equal(msg, msg2) and it == it2
            if dj2 is None:
                assert info[1] == 0
            else:
                assert info[1] == 1 and info[2] == dj2
                seen.add(("ml", dj2))
            if dj2 in (None, 0):
                assert np.array_equal(msg, cw)
    assert seen, "ML stage never exercised by this parameter set"


def test_decoder_matches_literal_code_a(oracle, code_a):
    """A few frames of the real (2040,1530) code incl. one that needs the ML stage (uniform 21%)."""
    H = code_a.dense()
    oc = oracle.OracleCode(code_a)
    src = synth.source(5, 0, 3, code_a.k, 1)[:, :, 0]
    for f, per in enumerate((0.10, 0.18, 0.215)):
        cw = oc.encode(src[f]).astype(np.int16)
        era = synth.erasures_uniform(77, f, 1, code_a.n, per)[0].astype(bool)
        recv = cw.copy()
        recv[era] = -1
        msg, it, info, rc = oc.decode(recv)
        msg2, it2, dj2 = ml.hybridml_nonbinary_decode(recv.astype(int), H.astype(int))
        assert rc == 0 and np.array_equal(msg, msg2) and it == it2
        if dj2 in (None, 0):
            assert np.array_equal(msg, cw)


def test_packet_decode_equals_lanewise_scalar_decode(oracle):
    """each byte lane of an S>1 decode equals an S=1 decode of that lane."""
    H = small_code(3)
    c = codes.from_dense(H, 36)
    oc = oracle.OracleCode(c)
    rng = np.random.default_rng(5)
    S = 8
    for per in (0.15, 0.35, 0.42):
        src = rng.integers(0, 256, size=(36, S)).astype(np.uint8)
        cw = oc.encode(src)
        era = (rng.random(60) <= per).astype(np.uint8)
        if era.sum() > H.shape[0]:
            era[np.nonzero(era)[0][H.shape[0]:]] = 0
        sym = cw.copy()
        sym[era.astype(bool)] = 0xAA  # payload of erased symbols must not matter
        out, oe, it, info, rc = oc.decode_packets(sym, era)
        for l in range(S):
            recv = cw[:, l].astype(np.int16)
            recv[era.astype(bool)] = -1
            msg, it1, info1, rc1 = oc.decode(recv)
            assert it1 == it and np.array_equal(info1, info)
            assert np.array_equal(msg.astype(np.uint8), out[:, l])


def test_decoder_edge_cases(oracle, code_a):
    oc = oracle.OracleCode(code_a)
    cw = oc.encode(synth.source(1, 0, 1, code_a.k, 1)[0, :, 0]).astype(np.int16)
    # no erasure: one sweep still runs
    msg, it, info, rc = oc.decode(cw)
    assert it == 1 and np.array_equal(msg, cw) and info[1] == 0
    # a single erasure
    recv = cw.copy(); recv[17] = -1
    msg, it, info, rc = oc.decode(recv)
    assert np.array_equal(msg, cw) and it <= 2
    # all parity erased: one in-order sweep re-encodes (triangle form)
    recv = cw.copy(); recv[code_a.k:] = -1
    msg, it, info, rc = oc.decode(recv)
    assert np.array_equal(msg, cw) and it == 1
    # more residual erasures than checks: Matlab would index past rhs -> rc -2, MP result returned
    recv = cw.copy(); recv[:600] = -1
    msg, it, info, rc = oc.decode(recv)
    assert rc == -2 and it == 10 and (msg == -1).sum() == info[0] > code_a.m
    # ML off: residual stays -1
    era = synth.erasures_uniform(3, 0, 1, code_a.n, 0.22)[0].astype(bool)
    recv = cw.copy(); recv[era] = -1
    msg, it, info, rc = oc.decode(recv, do_ml=0)
    assert info[1] == 0 and (msg == -1).sum() == info[0]


# ---------------------------------------------------------------- a10: binary siblings
def test_binary_decoders(oracle, code_a):
    cb = code_a.binary()
    oc = oracle.OracleCode(cb)
    Hb = cb.dense()
    rng = np.random.default_rng(9)
    src = rng.integers(0, 2, size=cb.k).astype(np.uint8)
    cw = oc.encode(src).astype(np.int16)
    assert set(np.unique(cw)) <= {0, 1}
    for per in (9 / 64, 0.215):
        era = synth.erasures_uniform(123, 0, 1, cb.n, per)[0].astype(bool)
        recv = cw.copy(); recv[era] = -1
        msg, it = oc.binary_mp(recv, itenum=50)
        msg2, it2 = ml.binary_mp_decode(recv.astype(int), Hb.astype(int), itenum=50)
        assert np.array_equal(msg, msg2) and it == it2
        # GF(2) hybrid == GF(256) hybrid with all-one coefficients on 0/1 data (same control flow)
        mh, ith, infoh, rch = oc.binary_hybrid(recv)
        mg, itg, infog, rcg = oc.decode(recv)
        assert np.array_equal(mh, mg) and ith == itg and np.array_equal(infoh, infog)
        if infog[2] == 0:
            assert np.array_equal(mg, cw)


def test_fpga_perf_tests_decoder_restatements_agree(oracle, code_a):
    """a10, FPGA flavour: the C restatement of OpenCL/device/ldpc_erasure_decoder_perf_tests.cl:56-236 against a second
    one written independently (tests/matlab_literal.py), on the reference's own (2040,1530) rows, with packet payloads:
    64-bit words of a valid codeword lane by lane, erased packets stored as zeros (the kernel's assumption 2).
    Includes frames in which the kernel's stop rule (num_current_correct == k, :205) fires before the systematic part
    is complete -- the restatements must agree on those too."""
    cb = code_a.binary()
    oc = oracle.OracleCode(cb)
    n, k = cb.n, cb.k
    vlist = [[int(cb.row_ptr[r + 1] - cb.row_ptr[r])] + [int(c) + 1 for c in cb.cols[cb.row_ptr[r]:cb.row_ptr[r + 1]]] for r in range(n - k)]
    rng = np.random.default_rng(77)
    # a 64-bit payload word per symbol: 64 independent binary codewords side by side
    bits = np.stack([oc.encode(rng.integers(0, 2, size=k).astype(np.uint8)) for _ in range(64)]).astype(np.uint64)
    payload = (bits << np.arange(64, dtype=np.uint64)[:, None]).sum(axis=0).astype(np.uint64)
    era_all = synth.fpga_erasures(5, 9, 40, n)
    premature = 0
    for f in range(40):
        er = era_all[f]
        sym = payload.copy()
        sym[er.astype(bool)] = 0
        for num_iter in (3, 50):   # the 50-iteration result is the one compared with the in-order decoder below
            left, it, er_out, pl_out = oc.fpga_perf_decoder(er, num_iter, sym)
            left2, it2, er2, pl2 = ml.fpga_perf_decoder(vlist, n, k, er, [int(x) for x in sym], num_iter)
            assert (left, it) == (left2, it2) and np.array_equal(er_out, np.array(er2, dtype=np.uint8))
            assert [int(x) for x in pl_out] == pl2
            known = er_out == 0
            assert np.array_equal(pl_out[known], payload[known])   # every recovered packet is the transmitted one
        recv = np.zeros(n, dtype=np.int16); recv[er.astype(bool)] = -1
        msg, _ = oc.binary_mp(recv, itenum=50)
        premature += int(left > 0 and not (msg[:k] == -1).any())
    assert premature > 0   # the rule does stop early on this stream (documented in DESIGN.md)


# ---------------------------------------------------------------- a5: Reed-Solomon
def test_rs_7_5_round_trip_like_reference_script(oracle):
    """Matlab/Test_My_RS_Decode.m:45-58 with (n,k) = (7,5): every k-subset of received positions."""
    import itertools
    n, k = 7, 5
    G = oracle.rs_generator(n, k)
    assert np.array_equal(G, ml.rs_generator(n, k))
    assert np.array_equal(G[:, :k], np.eye(k, dtype=np.uint8))  # systematic
    rng = np.random.default_rng(1)
    for ind in itertools.combinations(range(n), k):
        src = rng.integers(0, 256, size=k).astype(np.uint8)
        cw = oracle.rs_encode(G, src)
        ind = np.array(ind, dtype=np.uint16)
        msg, rc = oracle.rs_decode(G, ind, cw[ind])
        assert rc == 0 and np.array_equal(msg, src)
        assert np.array_equal(ml.rs_decode((ind + 1).tolist(), cw[ind].astype(int), n, k, G.astype(int)), src)


@pytest.mark.parametrize("n,k", [(255, 223), (255, 192), (250, 125)])
def test_rs_full_size_round_trip(oracle, n, k):
    G = oracle.rs_generator(n, k)
    assert np.array_equal(G[:, :k], np.eye(k, dtype=np.uint8))
    rng = np.random.default_rng(n + k)
    for trial in range(4):
        src = rng.integers(0, 256, size=k).astype(np.uint8)
        cw = oracle.rs_encode(G, src)
        nerase = [0, 1, (n - k) // 2, n - k][trial]
        keep = np.sort(rng.permutation(n)[: n - nerase])[:k].astype(np.uint16)
        msg, rc = oracle.rs_decode(G, keep, cw[keep])
        assert rc == 0 and np.array_equal(msg, src)


# ---------------------------------------------------------------- synthesised (4080,3060) code (NOT the authors')
def test_synthesised_code_c_structure(oracle):
    """tools/hgen.cpp output: triangle form, no 4-cycles, degree profile close to
    Matlab/Hgen_irregularDegree_no6cycles_systematic_encoding.m:38-40, and it encodes/decodes."""
    import scipy.sparse as sp
    c = codes.load_builtin(3)
    assert (c.n, c.k, c.m) == (4080, 3060, 1020)
    assert np.array_equal(c.cols[c.row_ptr[1:] - 1], c.k + np.arange(c.m))
    rd = np.diff(c.row_ptr.astype(np.int64))
    assert rd[-1] == 2 and set(np.unique(rd[:-1])) <= {13, 14, 15, 16}
    assert np.bincount(c.cols, minlength=c.n).max() <= 16
    H = sp.csr_matrix((np.ones(c.nnz, dtype=np.int32), c.cols.astype(np.int64), c.row_ptr.astype(np.int64)), shape=(c.m, c.n))
    G = (H @ H.T).tolil()
    G.setdiag(0)
    assert G.tocsr().max() <= 1  # two checks never share two symbols: no 4-cycles
    oc = oracle.OracleCode(c)
    src = synth.source(3, 0, 2, c.k, 1)[:, :, 0]
    cw = np.stack([oc.encode(s) for s in src])
    era = synth.erasures_uniform(4, 0, 2, c.n, 0.12)
    out, sw, res, st = oc.decode_batch_s1(cw, era)
    assert np.array_equal(out, cw) and (st == 0).all()


# ---------------------------------------------------------------- threefry (FPGA source kernel's generator)
def test_threefry4x32_20_known_answers(oracle):
    """Random123 known-answer vectors for threefry4x32, 20 rounds (the generator of
    OpenCL/device/ldpc_erasure_decoder_top.cl:74-97).  Own implementation, include/ldpc_erasure_amd_synth.h."""
    z = [0, 0, 0, 0]
    f = [0xFFFFFFFF] * 4
    assert [int(x) for x in oracle.threefry4x32_20(z, z)] == [0x9C6CA96A, 0xE17EAE66, 0xFC10ECD4, 0x5256A7D8]
    assert [int(x) for x in oracle.threefry4x32_20(f, f)] == [0x2A881696, 0x57012287, 0xF6C7446E, 0xA16A6732]
    ctr = np.array([z, f, [1, 2, 3, 4]], dtype=np.uint32)
    for key in (z, f, [1, 4242, 0, 0]):
        got = synth.threefry4x32_20(ctr, key)
        for i in range(3):
            assert np.array_equal(got[i], oracle.threefry4x32_20(ctr[i], key))
    era = oracle.fpga_data_in_erasures(4242, 9, 50, 2040)
    assert np.array_equal(era, synth.fpga_erasures(4242, 9, 50, 2040))
    assert abs(era.mean() - 9 / 64) < 0.01


def test_stat_helpers_two_sample_and_rounding():
    """tests/stat_helpers.py (used by the Table I reproduction): sanity of the conditional two-sample test."""
    from stat_helpers import consistent_with_rate, consistent_with_reported, two_sample_pvalue
    assert two_sample_pvalue(0, 10**6, 0, 10**6) == 1.0
    assert two_sample_pvalue(130, 10**6, 130, 10**6) > 0.9
    assert two_sample_pvalue(130, 10**6, 400, 10**6) < 1e-10
    assert consistent_with_reported(20400, 10**6, 0.015, 0.025, 10**6)[0]
    assert not consistent_with_reported(74000, 10**6, 0.015, 0.025, 10**6)[0]
    assert consistent_with_reported(3, 10**7, 0.0, 0.5e-6, 10**7)[0]
    assert not consistent_with_reported(60, 10**7, 0.0, 0.5e-6, 10**7)[0]
    assert consistent_with_rate(7223, 8 * 10**6, 9.028e-4)[0]
    assert not consistent_with_rate(9000, 8 * 10**6, 9.028e-4)[0]


# ---------------------------------------------------------------- cycle checker (generator + cycle checker)
def _run_hcycles(tmp_path, path, *flags):
    import subprocess
    exe = tmp_path / "hcycles"
    if not exe.exists():
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tools", "hcycles.cpp")])
    out = subprocess.run([str(exe), path, *flags], capture_output=True, text=True, check=True).stdout.splitlines()
    kv = {a: int(b) for a, b in (tok.split("=") for tok in out[0].split())}
    roots = [int(x) for x in out[1].split()[1:]] if len(out) > 1 else None
    return kv, roots


def _write_csr(path, n, k, rows):
    import struct
    row_ptr, cols = [0], []
    for r in rows:
        cols.extend(sorted(r))
        row_ptr.append(len(cols))
    with open(path, "wb") as f:
        f.write(b"LDPCCSR1" + struct.pack("<4I", n, k, n - k, len(cols)))
        f.write(np.asarray(row_ptr, dtype="<u4").tobytes() + np.asarray(cols, dtype="<u2").tobytes())


def test_cycle_checker_on_hand_made_graphs(tmp_path):
    """tools/hcycles.cpp (behaviour of Matlab/Hcyclefinder.m, Cycle_Finder_length4_fromroot.m, Cycle_Finder_length6.m) on
    graphs whose cycles are known by construction."""
    p = str(tmp_path / "c4.bin")
    _write_csr(p, 5, 3, [[0, 1, 3], [0, 1, 4]])                    # checks 0 and 1 share variables 0 and 1: one 4-cycle
    kv, roots = _run_hcycles(tmp_path, p, "--roots")
    assert kv["roots4"] == 2 and kv["pairs4"] == 2 and kv["girth_at_least"] == 4
    # like the reference's tree search, a root hanging off the 4-cycle (variables 3 and 4) sees it as a repeated check in
    # its check tier 2, i.e. reports a 6-cycle (Hcyclefinder.m:110-121 does not require the cycle to pass through the root)
    assert roots == [0, 1, 3, 4]
    p = str(tmp_path / "c6.bin")
    _write_csr(p, 6, 3, [[0, 1, 3], [1, 2, 4], [0, 2, 5]])         # triangle v0-c0-v1-c1-v2-c2-v0: one 6-cycle
    kv, roots = _run_hcycles(tmp_path, p, "--roots")
    assert kv["roots4"] == 0 and kv["roots6"] == 3 and kv["pairs6"] == 3 and kv["girth_at_least"] == 6 and roots == [0, 1, 2]
    p = str(tmp_path / "tree.bin")
    _write_csr(p, 7, 4, [[0, 1, 4], [1, 2, 5], [2, 3, 6]])         # a path: no cycle at all
    kv, _ = _run_hcycles(tmp_path, p)
    assert (kv["roots4"], kv["roots6"], kv["roots8"]) == (0, 0, 0) and kv["girth_at_least"] == 10


def test_cycle_census_of_the_shipped_matrices(tmp_path):
    """The survey-verified girth facts of the reference's three matrices (SURVEY.md section 8 preamble: no 4-cycles; 41 / 5 / 0
    variable roots that see a residual 6-cycle for codes A / B / D, left by the degree-1 clean-up of
    Hgen_irregularDegree_no6cycles_systematic_encoding.m:217-224), and the census of the synthesised (4080,3060) code."""
    want = {0: (0, 0), 1: (0, 41), 2: (0, 5)}
    for ci, (r4, r6) in want.items():
        kv, _ = _run_hcycles(tmp_path, codes.builtin_path(ci), "--no8")
        assert (kv["roots4"], kv["roots6"]) == (r4, r6), (ci, kv)
    if codes.have_builtin(3):
        kv, roots = _run_hcycles(tmp_path, codes.builtin_path(3), "--roots", "--no8")
        assert kv["roots4"] == 0                      # girth >= 6 everywhere ...
        assert kv["roots6"] <= 0.02 * kv["n"]         # ... and 6-cycles only at a handful of roots (41 of 4080)
        assert all(r >= 0 for r in roots)
