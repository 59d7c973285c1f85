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
