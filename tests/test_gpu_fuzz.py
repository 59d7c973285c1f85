"""Seeded fuzz over code shapes: random triangle-form codes of odd sizes (every degree bucket, heavy columns that force
the gather kernel, fewer checks than a wavefront, n of no convenient multiple), S = 1 and packets, erasure rates from
'nothing to do' to 'ML stage, some rank-deficient' -- the GPU must equal the oracle bit for bit on every frame,
`iterations`, residual counts and status words included."""
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


def random_code(rng):
    n = int(rng.integers(12, 640))
    m = int(np.clip(rng.integers(max(2, n // 10), max(3, (2 * n) // 3)), 2, n - 1))
    k = n - m
    rowdeg = int(rng.integers(2, 25))
    heavy = rng.random() < 0.25      # a few source columns appear in most checks: column degree > 16
    H = np.zeros((m, n), dtype=np.uint8)
    for i in range(m):
        avail = k + i
        c = rng.choice(avail, size=min(rowdeg - 1, avail), replace=False)
        H[i, c] = rng.integers(1, 256, size=c.size)
        if heavy and k >= 2:
            H[i, :2] = rng.integers(1, 256, size=2)
        H[i, k + i] = rng.integers(1, 256)
    return codes.from_dense(H, k)


@pytest.mark.parametrize("seed", range(int(os.environ.get("LDPC_FUZZ_SEEDS", "12"))))   # LDPC_FUZZ_SEEDS=200: a longer soak
def test_random_codes_match_the_oracle(ctx, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    for _ in range(3):
        code = random_code(rng)
        n, k = code.n, code.k
        if int(np.diff(code.row_ptr.astype(np.int64)).max()) > 24:
            with pytest.raises(api.LdpcAmdError):   # heavy columns on top of 24 random ones: beyond the kernels' row degree
                ctx.register_code(code)
            continue
        h = ctx.register_code(code)
        oc = oracle.OracleCode(code)
        F = 6
        for S in (1, int(rng.choice([16, 48, 80, 256, 528]))):
            src = synth.source(int(rng.integers(1 << 30)), 0, F, k, S)
            cw = ctx.encode(h, src[:, :, 0] if S == 1 else src).reshape(F, n, S)
            assert np.array_equal(cw[0], oc.encode(src[0, :, 0] if S == 1 else src[0]).reshape(n, S))
            rate = (n - k) / n
            pers = rng.uniform(0.0, 1.25 * rate, size=F)
            erased = np.stack([(rng.random(n) < p).astype(np.uint8) for p in pers])
            sym = cw.copy()
            if rng.random() < 0.5:
                sym[F // 2:] = rng.integers(0, 256, size=sym[F // 2:].shape, dtype=np.uint8)   # not codewords
            sym[erased.astype(bool)] = 0xEE
            max_sweeps = int(rng.choice([1, 3, 10, 50]))
            do_ml = int(rng.random() < 0.8)
            out, sw, res, st = ctx.decode(h, sym[:, :, 0] if S == 1 else sym, erased, max_sweeps=max_sweeps, do_ml=do_ml)
            out = out.reshape(F, n, S)
            for f in range(F):
                o_out, o_er, o_it, info, rc = oc.decode_packets(sym[f], erased[f], itenum=max_sweeps, do_ml=do_ml)
                want = 0 if info[0] == 0 else (3 if (rc == -2 or not info[1]) else (2 if info[2] else 1))
                ctxt = (seed, n, k, S, f, max_sweeps, do_ml, int(erased[f].sum()))
                assert sw[f] == o_it and res[f] == info[0] and st[f] == want, (ctxt, sw[f], o_it, res[f], info, st[f], want)
                if want != 3:
                    assert np.array_equal(out[f], o_out), (ctxt, want)
                else:
                    known = erased[f] == 0   # skipped / ML off: what was received or solved by the sweeps stays as it is
                    assert np.array_equal(out[f][known], sym[f][known]), ctxt
