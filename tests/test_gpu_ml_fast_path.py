"""The ML stage's fast path in packet mode (csrc/ml_pi.inc: peel on + inactivation, ML_PI knob) on the GPU, through the C-ABI:
  * codewords with erasures (bursty channel of BASELINE cfg 3): every mode -- exact only, verified fast path, unverified fast
    path; factorisation behind / beside the packet kernel -- returns the oracle's bytes, sweeps, residuals and status words,
    rank-deficient frames included;
  * received symbols that are NOT codewords: the verified fast path (the default) still equals the oracle on every frame -- its
    consistency test sends such frames to the exact elimination -- while the unverified one does not (that is what the test is
    for, and why ML_PI=2 is opt-in);
  * the (4000,2000) code, whose rows do not fit the fast path's LDS staging (read from global memory), and a schedule arena too
    small for the batch (frames deferred behind the packet kernel)."""
import numpy as np
import pytest

from ldpc_erasure_codes_amd import api, codes, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _reset(ctx):
    for k in ("ML_PI", "ML_OVERLAP", "ML_ARENA_WORDS", "ML_PI_IMAX"):
        ctx.configure("LDPC_AMD_" + k, None)


def _bursty_batch(ctx, h, code, F, S, seed):
    src = synth.source(seed, 0, F, code.k, S)
    cw = ctx.encode(h, src)
    era = synth.erasures_bursty(seed + 1, 0, 3 * F, code.n, 0.13, 0.8, 10.0)
    era = np.ascontiguousarray(era[era.sum(axis=1) < code.n - code.k][:F])
    assert era.shape[0] == F
    sym = cw.copy()
    sym[era.astype(bool)] = 0xA5
    return cw, sym, era


def _oracle_frames(oc, sym, era, frames):
    res = {}
    for f in frames:
        o_out, _, o_it, info, rc = oc.decode_packets(sym[f], era[f])
        want = 0 if info[0] == 0 else (3 if (rc == -2 or not info[1]) else (2 if info[2] else 1))
        res[f] = (o_out, o_it, int(info[0]), want)
    return res


def test_every_mode_equals_the_oracle_on_codewords(ctx, oracle, code_a):
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    oc = oracle.OracleCode(code_a)
    F, S = 160, 64
    cw, sym, era = _bursty_batch(ctx, h, code_a, F, S, 5150)
    _reset(ctx)
    try:
        ref = ctx.decode(h, sym, era)
        out0, sw0, res0, st0 = ref
        assert (st0 == 1).sum() >= 20 and (st0 == 2).sum() >= 1, np.bincount(st0)   # ML solved, and rank-deficient ones
        ml = np.flatnonzero(st0 >= 1)
        check = list(ml[:: max(1, len(ml) // 24)]) + list(np.flatnonzero(st0 == 2))[:4]
        for f, (o_out, o_it, o_res, want) in _oracle_frames(oc, sym, era, check).items():
            assert sw0[f] == o_it and res0[f] == o_res and st0[f] == want, f
            assert np.array_equal(out0[f], o_out), f
        assert np.array_equal(out0[st0 <= 1], cw[st0 <= 1])
        ms = ctx.ml_stats()   # the default run: most residual frames through the fast path, none flagged (the symbols are codewords)
        assert ms["residual_frames"] == int((st0 >= 1).sum()) and ms["flagged_frames"] == 0 and ms["deferred_frames"] == 0, ms
        assert ms["fast_path_frames"] >= 0.8 * ms["residual_frames"] and ms["fast_path_frames"] <= ms["residual_frames"] - int((st0 == 2).sum()), ms
        for pi in ("0", "1", "2"):
            for ov in ("0", "1", "2"):
                ctx.configure("LDPC_AMD_ML_PI", pi)
                ctx.configure("LDPC_AMD_ML_OVERLAP", ov)
                got = ctx.decode(h, sym, era)
                for a_, b_, what in zip(got, ref, ("out", "sweeps", "residual", "status")):
                    assert np.array_equal(a_, b_), (pi, ov, what)
                ms = ctx.ml_stats()
                assert (ms["fast_path_frames"] == 0) == (pi == "0") and ms["flagged_frames"] == 0, (pi, ov, ms)
    finally:
        _reset(ctx)


def test_symbols_that_are_not_codewords(ctx, oracle, code_a):
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    oc = oracle.OracleCode(code_a)
    F, S = 96, 32
    cw, sym, era = _bursty_batch(ctx, h, code_a, F, S, 6260)
    rng = np.random.default_rng(7)
    for f in range(F // 2, F):                      # second half: three received symbols of every frame corrupted
        known = np.flatnonzero(era[f] == 0)
        for j in rng.choice(known, size=3, replace=False):
            sym[f, j] ^= rng.integers(1, 256, size=S, dtype=np.uint8)
    _reset(ctx)
    try:
        ctx.configure("LDPC_AMD_ML_PI", "0")
        exact = ctx.decode(h, sym, era)
        ctx.configure("LDPC_AMD_ML_PI", "1")
        verified = ctx.decode(h, sym, era)
        ms = ctx.ml_stats()   # the corrupted frames the fast path solved are the ones its consistency test must catch
        n_bad_ml = int((exact[3][F // 2:] == 1).sum())
        assert ms["flagged_frames"] >= 0.8 * n_bad_ml and ms["flagged_frames"] <= n_bad_ml, (ms, n_bad_ml)
        ctx.configure("LDPC_AMD_ML_PI", "2")
        unverified = ctx.decode(h, sym, era)
        for a_, b_, what in zip(verified, exact, ("out", "sweeps", "residual", "status")):
            assert np.array_equal(a_, b_), what
        st = exact[3]
        ml_bad = [f for f in range(F // 2, F) if st[f] == 1]
        assert len(ml_bad) >= 8
        for f, (o_out, o_it, o_res, want) in _oracle_frames(oc, sym, era, ml_bad[:12]).items():
            assert verified[1][f] == o_it and verified[2][f] == o_res and verified[3][f] == want, f
            assert np.array_equal(verified[0][f], o_out), f
        # the unverified mode is only for inputs known to be codewords: on these frames it returns other bytes
        differs = [f for f in ml_bad if not np.array_equal(unverified[0][f], exact[0][f])]
        assert differs, "the corrupted frames should make the unverified fast path visible"
        good = [f for f in range(F // 2) if st[f] <= 1]
        assert np.array_equal(unverified[0][good], exact[0][good])
    finally:
        _reset(ctx)


def test_code_b_and_a_small_arena(ctx, oracle):
    code_b = codes.load_builtin(2, codes.DEFAULT_COEF_SEED[2])
    h = ctx.load_builtin_code(2, codes.DEFAULT_COEF_SEED[2])
    oc = oracle.OracleCode(code_b)
    F, S = 40, 16
    src = synth.source(99, 0, F, code_b.k, S)
    cw = ctx.encode(h, src)
    pers = np.linspace(0.40, 0.52, F)               # around the (4000,2000) code's thresholds: sweeps cap, ML, rank deficiency
    era = np.concatenate([synth.erasures_uniform(200 + i, i, 1, code_b.n, float(p)) for i, p in enumerate(pers)])
    sym = cw.copy()
    sym[era.astype(bool)] = 0x3C
    _reset(ctx)
    try:
        ref = ctx.decode(h, sym, era)
        st = ref[3]
        assert (st >= 1).sum() >= 6, np.bincount(st)
        ml = list(np.flatnonzero(st >= 1))[:8]
        for f, (o_out, o_it, o_res, want) in _oracle_frames(oc, sym, era, ml).items():
            assert ref[1][f] == o_it and ref[2][f] == o_res and st[f] == want, f
            if want != 3:
                assert np.array_equal(ref[0][f], o_out), f
        for knobs in ({"ML_PI": "0"}, {"ML_ARENA_WORDS": "30000"}, {"ML_ARENA_WORDS": "30000", "ML_OVERLAP": "1"},
                      {"ML_ARENA_WORDS": "30000", "ML_PI": "2"}):
            _reset(ctx)
            for k, v in knobs.items():
                ctx.configure("LDPC_AMD_" + k, v)
            got = ctx.decode(h, sym, era)
            for a_, b_, what in zip(got, ref, ("out", "sweeps", "residual", "status")):
                assert np.array_equal(a_, b_), (knobs, what)
            if "ML_ARENA_WORDS" in knobs:
                assert ctx.ml_stats()["deferred_frames"] > 0, (knobs, ctx.ml_stats())   # the small arena did overflow
    finally:
        _reset(ctx)


# ---- round 4: ONE inconsistent byte lane at S = 1024 ------------------------------------------------------------
# At S = 1024 a frame is eight solve-kernel workgroups (128-byte slices), each of which runs the consistency test on its own
# slice and only ONE of which can see a single corrupted byte lane.  The expected set of flagged frames is computed exactly, not
# bounded: a frame is flagged iff, in some lane, the residual system H(touched checks, unknowns) x = rhs is INCONSISTENT
# (rank [A | rhs] > rank A), with rhs built from the lane's values after message passing (...Decoder.m:63-82).
def _gf_tables():
    import os
    import sys
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import pi_model
    return pi_model.MUL, pi_model.INV


def _lane_consistent(code, oc, lane_sym, era):
    """lane_sym uint8[n] (one byte lane of a frame), era uint8[n] -> True when the residual system of that lane is consistent."""
    MUL, INV = _gf_tables()
    recv = lane_sym.astype(np.int16)
    recv[era != 0] = -1
    msg, _, _, _ = oc.decode(recv, do_ml=0)            # the lane after the <= 10 sweeps of message passing
    unk = np.flatnonzero(msg < 0)
    assert unk.size
    col_of = {int(v): i for i, v in enumerate(unk)}
    rows_A, rhs = [], []
    for r in range(code.m):
        s, e = int(code.row_ptr[r]), int(code.row_ptr[r + 1])
        cs, hs = code.cols[s:e], code.coefs[s:e]
        mask = msg[cs] < 0
        if not mask.any():
            continue                                    # untouched check: an all-zero row of H(:, erased), never a pivot row
        a = np.zeros(unk.size, dtype=np.uint8)
        for c, h in zip(cs[mask], hs[mask]):
            a[col_of[int(c)]] = h
        b = 0
        for c, h in zip(cs[~mask], hs[~mask]):
            b ^= int(MUL[h, msg[c]])
        rows_A.append(a)
        rhs.append(b)
    M = np.concatenate([np.stack(rows_A), np.array(rhs, dtype=np.uint8)[:, None]], axis=1)
    T, E = M.shape[0], unk.size
    r = 0
    for c in range(E):
        piv = np.flatnonzero(M[r:, c])
        if piv.size == 0:
            continue
        p = r + int(piv[0])
        if p != r:
            M[[r, p]] = M[[p, r]]
        M[r] = MUL[INV[M[r, c]], M[r]]
        rows = np.flatnonzero(M[:, c])
        rows = rows[rows != r]
        M[rows] ^= MUL[M[rows, c][:, None], M[r][None, :]]
        r += 1
        if r == T:
            break
    return not M[r:, E].any()


def test_one_corrupted_byte_lane_at_S1024_is_flagged_exactly(ctx, oracle, code_a):
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    oc = oracle.OracleCode(code_a)
    F, S = 40, 1024
    cw, sym, era = _bursty_batch(ctx, h, code_a, F, S, 7370)
    _reset(ctx)
    try:
        clean = ctx.decode(h, sym, era)
        st = clean[3]
        ml1 = [int(f) for f in np.flatnonzero(st == 1)]
        ms = ctx.ml_stats()
        assert len(ml1) >= 8, np.bincount(st)
        # every status-1 frame of this batch went through the fast path (so "flagged" below is decided by consistency alone)
        assert ms["fast_path_frames"] == len(ml1) and ms["flagged_frames"] == 0, (ms, len(ml1))
        lanes = [3, 5 * 128 + 17, S - 1]                # a lane of slice 0, of slice 5, the last lane of the last slice
        rng = np.random.default_rng(11)
        colrows = [[] for _ in range(code_a.n)]
        for r in range(code_a.m):
            for c in code_a.cols[int(code_a.row_ptr[r]):int(code_a.row_ptr[r + 1])]:
                colrows[int(c)].append(r)
        bad = sym.copy()
        expect_flag, quiet_done = set(), False
        touched_frames = []
        for i, f in enumerate(ml1[:9]):
            lane = lanes[i % 3]
            known = np.flatnonzero(era[f] == 0)
            want_quiet = (i % 3 == 2) and not quiet_done   # one frame whose corruption no equation of the system sees
            chosen = None
            order = rng.permutation(known)[:40]
            if want_quiet:   # symbols none of whose checks has an erased neighbour: no equation of the frame sees them (...Decoder.m:74-82)
                rows_er = set()
                for e_ in np.flatnonzero(era[f]):
                    rows_er.update(colrows[int(e_)])
                order = [j for j in known if not (set(colrows[int(j)]) & rows_er)] + list(order)
            for j in order:
                trial = bad[f, :, lane].copy()
                trial[j] ^= 0x4D
                cons = _lane_consistent(code_a, oc, trial, era[f])
                if cons == want_quiet:
                    chosen = int(j)
                    break
            if chosen is None:
                assert want_quiet, f
                j = int(rng.choice(known))              # (no invisible symbol in this frame: corrupt visibly instead)
                trial = bad[f, :, lane].copy()
                trial[j] ^= 0x4D
                cons = _lane_consistent(code_a, oc, trial, era[f])
                chosen = j
            bad[f, chosen, lane] ^= 0x4D
            if not cons:
                expect_flag.add(f)
            else:
                quiet_done = True
            touched_frames.append(f)
        assert expect_flag and quiet_done, (expect_flag, quiet_done)
        ctx.configure("LDPC_AMD_ML_PI", "0")
        exact = ctx.decode(h, bad, era)
        ctx.configure("LDPC_AMD_ML_PI", "1")
        verified = ctx.decode(h, bad, era)
        ms = ctx.ml_stats()
        for a_, b_, what in zip(verified, exact, ("out", "sweeps", "residual", "status")):
            assert np.array_equal(a_, b_), what
        assert ms["flagged_frames"] == len(expect_flag), (ms, sorted(expect_flag))
        for f in touched_frames:                        # every lane of the corrupted frames against the oracle's packet decode
            o_out, _, o_it, info, rc = oc.decode_packets(bad[f], era[f])
            assert verified[1][f] == o_it and verified[2][f] == info[0], f
            assert np.array_equal(verified[0][f], o_out), f
        # frames that were not touched decode as before
        others = [f for f in range(F) if f not in touched_frames]
        assert np.array_equal(verified[0][others], clean[0][others])
    finally:
        _reset(ctx)


def test_adaptive_skip_lags_one_batch_and_keeps_the_bytes(oracle, code_a):
    """ML_PI_ADAPTIVE (default on): after a packet batch with NO residual frame the next batch skips the fast path's launches -- its
    residual frames are all factored exactly (fast_path_frames == 0) --, the batch after that has the fast path again; the bytes
    are the same in all three states."""
    with api.Context(0) as c:
        h = c.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
        oc = oracle.OracleCode(code_a)
        F, S = 32, 64
        cw, sym, era = _bursty_batch(c, h, code_a, F, S, 8480)
        clean_era = synth.erasures_uniform(8481, 0, F, code_a.n, 0.05)
        clean_sym = cw.copy()
        clean_sym[clean_era.astype(bool)] = 0
        first = c.decode(h, sym, era)                     # fresh context: nothing known yet -> fast path
        ms0 = c.ml_stats()
        assert ms0["residual_frames"] > 0 and ms0["fast_path_frames"] > 0, ms0
        out = c.decode(h, clean_sym, clean_era)           # a batch message passing completes
        c.synchronize()
        assert int(out[3].max()) == 0 and np.array_equal(out[0], cw)
        lagged = c.decode(h, sym, era)                    # quiet context: exact elimination for everything
        ms1 = c.ml_stats()
        again = c.decode(h, sym, era)
        ms2 = c.ml_stats()
        assert ms1["residual_frames"] == ms0["residual_frames"] and ms1["fast_path_frames"] == 0, ms1
        assert ms2["fast_path_frames"] == ms0["fast_path_frames"], (ms0, ms2)
        for got in (lagged, again):
            for a_, b_, what in zip(got, first, ("out", "sweeps", "residual", "status")):
                assert np.array_equal(a_, b_), what
        f = int(np.flatnonzero(first[3] == 1)[0])
        o_out, _, o_it, info, rc = oc.decode_packets(sym[f], era[f])
        assert np.array_equal(lagged[0][f], o_out) and lagged[1][f] == o_it
        c.configure("LDPC_AMD_ML_PI_ADAPTIVE", "0")       # switched off: no lag
        c.decode(h, clean_sym, clean_era)
        c.synchronize()
        c.decode(h, sym, era)
        assert c.ml_stats()["fast_path_frames"] == ms0["fast_path_frames"]
