"""Round 4, GPU, through the C-ABI.

* The encoder's static schedule collapsed offline (DevCode::encg_*, knobs ENC_GROUP / ENC_CAP): consecutive dependency levels
  of the parity triangle become groups whose steps pull the raw accumulators of their in-group ancestors with composite
  coefficients.  GF(256) products are associative and distributive, so the codeword must be the row-by-row encoder's
  (Matlab/ErasureCodes_NonBinaryLDPCSim.m:173-182, OpenCL/device/ldpc_erasure_encoder.cl:72-84) byte for byte: every built-in code
  and random triangle codes, every cap, against the oracle and against the level-by-level schedule."""
import numpy as np
import pytest

from ldpc_erasure_codes_amd import api, codes, synth

pytestmark = pytest.mark.gpu


def _random_triangle_code(rng, n, k, deg):
    """(n-k) x n, parity part lower triangular with a non-zero diagonal (row i ends in column k+i), random GF(256) coefficients;
    long dependency chains on purpose: every row takes its previous parity symbol with probability 0.8."""
    m = n - k
    row_ptr, cols, coefs = [0], [], []
    for r in range(m):
        c = set(rng.choice(k, size=min(deg, k), replace=False).tolist())
        if r > 0 and rng.random() < 0.8:
            c.add(k + r - 1)
        for j in rng.choice(max(r, 1), size=min(2, r), replace=False).tolist() if r > 1 else []:
            c.add(k + j)
        c = sorted(c) + [k + r]
        cols += c
        coefs += rng.integers(1, 256, size=len(c)).tolist()
        row_ptr.append(len(cols))
    return codes.Code(n, k, np.array(row_ptr, dtype=np.uint32), np.array(cols, dtype=np.uint16), np.array(coefs, dtype=np.uint8))


@pytest.mark.parametrize("code_ind", [1, 3])
def test_persistent_encoder_equals_the_one_item_kernel_and_the_oracle(oracle, code_ind):
    """ENC_PERSIST = 1 (the default: persistent workgroups that set the code's tables up once and take (frame, slice) items from a
    self-resetting device counter) against ENC_PERSIST = 0 (one workgroup per item) and the oracle's row-by-row encoder
    (Matlab/ErasureCodes_NonBinaryLDPCSim.m:173-182): batches smaller and larger than the grid, call after call on one context (the
    counter must come back to zero), grouped and plain static schedules, with an S = 1 encode and a decode in between."""
    if not codes.have_builtin(code_ind):
        pytest.skip("fixture of this code not present")
    code = codes.load_builtin(code_ind)
    oc = oracle.OracleCode(code)
    with api.Context(0) as ctx:
        h = ctx.load_builtin_code(code_ind, codes.DEFAULT_COEF_SEED[code_ind])
        for group in ("1", "0"):
            ctx.configure("ENC_GROUP", group)
            for S, F in ((1024, 3), (128, 1), (256, 70), (1024, 130), (128, 7)):
                src = synth.source(900 + S + F, 0, F, code.k, S)
                ctx.configure("ENC_PERSIST", "1")
                cw = ctx.encode(h, src)
                cw_again = ctx.encode(h, src)
                ctx.configure("ENC_PERSIST", "0")
                cw0 = ctx.encode(h, src)
                assert np.array_equal(cw, cw0), (code_ind, group, S, F)
                assert np.array_equal(cw_again, cw0), (code_ind, group, S, F)
                for f in (0, F // 2, F - 1):
                    assert np.array_equal(cw[f], oc.encode(src[f])), (code_ind, group, S, F, f)
            ctx.configure("ENC_PERSIST", None)
            # other kernels of the context in between, then the persistent encoder again
            s1 = synth.source(77, 0, 5, code.k, 1)[:, :, 0]
            c1 = ctx.encode(h, s1)
            assert np.array_equal(c1[0], oc.encode(s1[0][:, None])[:, 0])
            src = synth.source(901, 0, 4, code.k, 128)
            cw = ctx.encode(h, src)
            er = np.zeros((4, code.n), dtype=np.uint8)
            er[:, ::9] = 1
            rx = cw.copy()
            rx[er.astype(bool)] = 0
            res = ctx.decode(h, rx, er)
            out = res[0] if isinstance(res, tuple) else res.out
            assert np.array_equal(out, cw)
            assert np.array_equal(ctx.encode(h, src), cw)
        ctx.configure("ENC_GROUP", None)
        # more launches than the ring of item counters has entries: every counter must have reset itself
        src = synth.source(902, 0, 3, code.k, 128)
        ref = ctx.encode(h, src)
        for _ in range(70):
            assert np.array_equal(ctx.encode(h, src), ref)


@pytest.mark.parametrize("code_ind", [1, 3, 2, 0])
def test_grouped_encoder_equals_the_row_by_row_encoder(oracle, code_ind):
    if not codes.have_builtin(code_ind):
        pytest.skip("fixture of this code not present")
    code = codes.load_builtin(code_ind)
    oc = oracle.OracleCode(code)
    with api.Context(0) as ctx:
        h = ctx.load_builtin_code(code_ind, codes.DEFAULT_COEF_SEED[code_ind])
        info = ctx.encode_info(h)
        assert info["levels"] >= 20 and 0 < info["groups"] < info["levels"] / 2 and info["max_pull"] <= 8, info
        for S, F in ((1024, 6), (128, 9), (256, 5), (16, 4)):
            src = synth.source(400 + S, 0, F, code.k, S)
            ctx.configure("ENC_GROUP", "1")
            cw = ctx.encode(h, src)
            used = ctx.encode_info(h)["last_encode_grouped"]
            ctx.configure("ENC_GROUP", "0")
            cw0 = ctx.encode(h, src)
            assert ctx.encode_info(h)["last_encode_grouped"] == 0
            assert np.array_equal(cw, cw0), (code_ind, S)
            if code_ind in (1, 3) and S >= 128:                   # the two matrices north_star names must take the grouped schedule
                assert used == 1, (code_ind, S, used)             # ((4000,2000): m = 2000 accumulators leave its lists no LDS)
            for f in (0, F - 1):
                assert np.array_equal(cw[f], oc.encode(src[f])), (code_ind, S, f)
        ctx.configure("ENC_GROUP", None)
        for knobs in ({"ENC_B": "256"}, {"ENC_LIST": "1"}, {"SCATTER_NT": "0"}, {"SCATTER_DYN": "0"}):
            for kk, v in knobs.items():
                ctx.configure(kk, v)
            src = synth.source(77, 0, 3, code.k, 512)
            cw = ctx.encode(h, src)
            assert np.array_equal(cw[1], oc.encode(src[1])), (code_ind, knobs)
            for kk in knobs:
                ctx.configure(kk, None)


@pytest.mark.parametrize("cap", [0, 1, 2, 5, 16, 64])
def test_every_cap_and_random_triangle_codes(oracle, cap):
    rng = np.random.default_rng(1000 + cap)
    with api.Context(0) as ctx:
        ctx.configure("ENC_CAP", str(cap))      # read when a code is registered
        for (n, k, deg) in ((96, 40, 5), (700, 300, 9), (2040, 1530, 12)):
            code = _random_triangle_code(rng, n, k, deg)
            oc = oracle.OracleCode(code)
            h = ctx.register_code(code)
            info = ctx.encode_info(h)
            assert (info["groups"] == 0) == (cap == 0), info
            assert info["max_pull"] <= cap, info
            for S in (128, 1024):
                src = synth.source(9 + S, 0, 3, code.k, S)
                cw = ctx.encode(h, src)
                for f in range(3):
                    assert np.array_equal(cw[f], oc.encode(src[f])), (cap, n, S, f)
            src1 = synth.source(5, 0, 4, code.k, 1)[:, :, 0]          # the S = 1 encoder is the peel kernel: unaffected
            cw1 = ctx.encode(h, src1)
            assert np.array_equal(cw1[2], oc.encode(src1[2]))


# ---- the time-stamp relaxation (csrc/peel_relax.inc) against the serial per-solve loop and the oracle -------------------------------
def _wide_row_code(rng, n, k, deg):
    """Random code whose checks have up to `deg` neighbours (not triangular: decode only)."""
    m = n - k
    row_ptr, cols, coefs = [0], [], []
    for r in range(m):
        c = sorted(rng.choice(n, size=int(rng.integers(2, deg + 1)), replace=False).tolist())
        cols += c
        coefs += rng.integers(1, 256, size=len(c)).tolist()
        row_ptr.append(len(cols))
    return codes.Code(n, k, np.array(row_ptr, dtype=np.uint32), np.array(cols, dtype=np.uint16), np.array(coefs, dtype=np.uint8))


@pytest.mark.parametrize("S", [1, 64])
def test_relaxation_equals_the_serial_loop_and_the_oracle(oracle, S):
    """PEEL_RELAX = 1 (the default) and 0 must agree on every byte and every status word -- iterations, residual counts, status --
    for sweep caps on both sides of the relaxation's limits (62 sweeps, 16-bit keys), on codewords and on symbols that are not
    codewords, with the ML stage on and off; a sample of frames is compared with the oracle (...Decoder.m:21-59)."""
    rng = np.random.default_rng(40 + S)
    cases = [(1, 0.10), (1, 0.19), (1, 0.26), (2, 0.40), (2, 0.47), (0, 0.38), (0, 0.45)]
    with api.Context(0) as ctx:
        for code_ind, per in cases:
            code = codes.load_builtin(code_ind)
            oc = oracle.OracleCode(code)
            h = ctx.load_builtin_code(code_ind, codes.DEFAULT_COEF_SEED[code_ind])
            F = 24
            src = synth.source(7 * code_ind + 1, 0, F, code.k, S)
            cw = ctx.encode(h, src if S > 1 else src[:, :, 0])
            era = synth.erasures_uniform(50 + code_ind, int(per * 100), F, code.n, per)
            sym = cw.copy()
            sym[era.astype(bool)] = 0x77
            for f in range(0, F, 5):                       # every fifth frame: two received symbols corrupted (not a codeword)
                known = np.flatnonzero(era[f] == 0)
                for j in rng.choice(known, size=2, replace=False):
                    sym[f, j] ^= 0x21
            for sweeps, do_ml in ((1, 1), (3, 0), (10, 1), (50, 1), (62, 0), (63, 1), (200, 0)):
                ctx.configure("PEEL_RELAX", "1")
                a = ctx.decode(h, sym, era, max_sweeps=sweeps, do_ml=do_ml)
                used = ctx.profile_kernel_names()["peel"]
                ctx.configure("PEEL_RELAX", "0")
                b = ctx.decode(h, sym, era, max_sweeps=sweeps, do_ml=do_ml)
                ctx.configure("PEEL_RELAX", None)
                for x, y, what in zip(a, b, ("out", "sweeps", "residual", "status")):
                    assert np.array_equal(x, y), (code_ind, per, sweeps, do_ml, what)
                logm = int(np.ceil(np.log2(max(64, -(-code.m // 64) * 64))))
                fits = sweeps <= 62 and ((sweeps + 1) << logm) <= 65535
                assert ("relax" in used) == fits, (code_ind, sweeps, used)      # the fall-back is taken exactly when the keys do not fit
                if sweeps == 10 and do_ml == 1:
                    for f in (0, 5, F - 1):
                        if S == 1:
                            o = oc.decode_batch_s1(sym[f:f + 1], era[f:f + 1])
                            assert np.array_equal(a[0][f], o[0][0]) and a[1][f] == o[1][0] and a[2][f] == o[2][0] and a[3][f] == o[3][0], (code_ind, per, f)
                        else:
                            o, _, it, info, rc = oc.decode_packets(sym[f], era[f])
                            assert a[1][f] == it and a[2][f] == info[0], (code_ind, per, f)
                            if not (rc == -2 or not info[1]):
                                assert np.array_equal(a[0][f], o), (code_ind, per, f)
        # a code whose checks are wider than the relaxation's 16 neighbours: the serial kernel takes it, same results as the oracle
        code = _wide_row_code(rng, 600, 300, 22)
        oc = oracle.OracleCode(code)
        h = ctx.register_code(code)
        symw = rng.integers(0, 256, size=(6, code.n) if S == 1 else (6, code.n, S), dtype=np.uint8)
        eraw = synth.erasures_uniform(3, 0, 6, code.n, 0.08)
        out = ctx.decode(h, symw, eraw, max_sweeps=10, do_ml=0)
        assert "relax" not in ctx.profile_kernel_names()["peel"]
        for f in range(6):
            if S == 1:
                o = oc.decode_batch_s1(symw[f:f + 1], eraw[f:f + 1], do_ml=0)
                assert np.array_equal(out[0][f], o[0][0]) and out[1][f] == o[1][0] and out[2][f] == o[2][0]
            else:
                o, _, it, info, rc = oc.decode_packets(symw[f], eraw[f], do_ml=0)
                assert np.array_equal(out[0][f], o) and out[1][f] == it and out[2][f] == info[0]


# ---- tier 2 of the packet kernel: several pieces of a frame per work item (SCATTER_T2P), the frame's set-up shared ---------------------
def test_tier2_pieces_per_item_equal_one_piece_per_item_and_the_oracle(oracle):
    """Bursty frames of BASELINE cfg 3 at S = 1024 (most have more steps than tier 1 holds; some reach the ML stage, some stay
    rank deficient): 1, 2 and 4 pieces of a frame per tier-2 work item -- the later pieces find the frame's tables in place -- must
    return the same bytes and status words, with plain and paired levels, and the oracle's on spot-checked
    frames (Matlab/My_LDPC_HybridML_NonBinary_Erasure_Decoder.m:13-129)."""
    code = codes.load_builtin(1)
    oc = oracle.OracleCode(code)
    with api.Context(0) as ctx:
        h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
        F, S = 40, 1024
        src = synth.source(4242, 0, F, code.k, S)
        cw = ctx.encode(h, src)
        era = synth.erasures_bursty(4243, 0, 3 * F, code.n, 0.13, 0.8, 10.0)
        era = np.ascontiguousarray(era[era.sum(axis=1) < code.n - code.k][:F])
        assert era.shape[0] == F and (era.sum(axis=1) > 300).sum() >= 10
        sym = cw.copy()
        sym[era.astype(bool)] = 0x5A
        ctx.configure("SCATTER_T2P_FORCE", "1")
        try:
            ref = None
            for pairs in ("1", "0"):
                ctx.configure("SCATTER_PAIRS", pairs)
                for tp in ("1", "2", "4"):
                    ctx.configure("SCATTER_T2P", tp)
                    out, sw, res, st = ctx.decode(h, sym, era)
                    if ref is None:
                        ref = (out, sw, res, st)
                        assert (st >= 1).sum() >= 3, np.bincount(st)
                        for f in (0, 7, F - 1, int(np.flatnonzero(st >= 1)[0])):
                            o_out, _, o_it, info, rc = oc.decode_packets(sym[f], era[f])
                            assert np.array_equal(out[f], o_out) and sw[f] == o_it, f
                    else:
                        assert np.array_equal(out, ref[0]) and np.array_equal(st, ref[3]) and np.array_equal(sw, ref[1]), (pairs, tp)
            for tp in ("2", "4"):   # list modes of the stream rebuild their row list per piece: they keep one piece per item
                ctx.configure("SCATTER_T2P", tp)
                for dyn in ("0", "2", "3", "4"):
                    ctx.configure("SCATTER_DYN", dyn)
                    out, sw, res, st = ctx.decode(h, sym, era)
                    assert np.array_equal(out, ref[0]) and np.array_equal(st, ref[3]), (tp, dyn)
                ctx.configure("SCATTER_DYN", None)
        finally:
            for k_ in ("SCATTER_T2P_FORCE", "SCATTER_T2P", "SCATTER_PAIRS", "SCATTER_DYN"):
                ctx.configure(k_, None)


# ---- long S = 1 batches: frames per launch (CHUNK_S1) ------------------------------------------------------------------------------
def test_long_s1_batch_is_the_same_in_one_launch_and_in_chunks(oracle):
    """Frames are independent (one decoder call per frame in the reference, Matlab/ErasureCodes_NonBinaryLDPCSim.m:218), so a long
    S = 1 batch decoded in one launch, in chunks with a ragged last one (CHUNK_S1 = 1024, 2048) and frame by frame by the oracle
    must agree on every byte and status word -- with frames in the batch that reach the ML stage and that it cannot finish."""
    code = codes.load_builtin(1)
    oc = oracle.OracleCode(code)
    F = 5000
    with api.Context(0) as ctx:
        h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
        src = synth.source(77, 0, F, code.k, 1)
        cw = ctx.encode(h, src[:, :, 0])
        era = synth.erasures_uniform(78, 0, F, code.n, 0.10)
        era[1000:1400] = synth.erasures_uniform(79, 0, 400, code.n, 0.21)     # frames for the ML stage, across the 1024-frame border
        era[4990:] = synth.erasures_uniform(80, 0, 10, code.n, 0.26)          # more erasures than checks: cannot be decoded
        sym = cw.copy()
        sym[era.astype(bool)] = 0x3C
        ctx.configure("CHUNK_S1", "1048576")
        one = ctx.decode(h, sym, era)
        assert set(one[3].tolist()) >= {0, 1} and one[3].max() >= 2
        for chunk in ("1024", "2048", None):
            ctx.configure("CHUNK_S1", chunk)
            got = ctx.decode(h, sym, era)
            for x, y, what in zip(one, got, ("out", "sweeps", "residual", "status")):
                assert np.array_equal(x, y), (chunk, what)
        for f in (0, 1023, 1024, 1100, 2047, 2048, 4095, 4096, 4995, F - 1):
            o = oc.decode_batch_s1(sym[f:f + 1], era[f:f + 1])
            assert np.array_equal(one[0][f], o[0][0]) and one[1][f] == o[1][0] and one[2][f] == o[2][0] and one[3][f] == o[3][0], f
        with pytest.raises(api.LdpcAmdError):
            ctx.configure("CHUNK_S1", "100")
