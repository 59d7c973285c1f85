"""Committed golden vectors (tests/golden/ldpc_golden_vectors.npz, made by tools/make_golden_vectors.py from the
CPU oracle -- the reference ships no decoder I/O vectors).  The CPU test keeps the oracle from drifting; the GPU
test checks the HIP path on the GPU box against the same bytes without needing anything else."""
import os

import numpy as np
import pytest

from ldpc_erasure_codes_amd import codes

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ldpc_golden_vectors.npz"))


def _era(bits, n):
    return np.unpackbits(bits, axis=1)[:, :n]


def test_oracle_reproduces_golden_vectors(oracle):
    code = codes.load_builtin(int(G["code_ind"]), int(G["coef_seed"]))
    oc = oracle.OracleCode(code)
    era = _era(G["erased_bits"], code.n)
    out, sw, res, st = oc.decode_batch_s1(G["sym"], era)
    assert np.array_equal(out, G["out"]) and np.array_equal(sw, G["sweeps"])
    assert np.array_equal(res, G["residual"]) and np.array_equal(st, G["status"])
    ok = np.isin(G["status"], (0, 1))
    assert np.array_equal(G["out"][ok], G["codeword"][ok])  # decodable frames return the codeword
    assert set(np.unique(G["status"])) == {0, 1, 2, 3}
    pera = _era(G["p_erased_bits"], code.n)
    for f in range(pera.shape[0]):
        o, _, it, info, rc = oc.decode_packets(G["p_sym"][f], pera[f])
        assert np.array_equal(o, G["p_out"][f]) and it == G["p_sweeps"][f] and info[0] == G["p_residual"][f]
    g = oracle.rs_generator(int(G["rs_n"]), int(G["rs_k"]))
    for b in range(G["rs_idx"].shape[0]):
        msg, rc = oracle.rs_decode(g, G["rs_idx"][b], G["rs_val"][b])
        assert rc == 0 and np.array_equal(msg, G["rs_msg"][b])


@pytest.mark.gpu
def test_gpu_reproduces_golden_vectors():
    from ldpc_erasure_codes_amd import api
    with api.Context(0) as ctx:
        h = ctx.load_builtin_code(int(G["code_ind"]), int(G["coef_seed"]))
        n, k, _ = ctx.code_info(h)
        era = np.ascontiguousarray(_era(G["erased_bits"], n))
        out, sw, res, st = ctx.decode(h, G["sym"], era)
        assert np.array_equal(out, G["out"]) and np.array_equal(sw, G["sweeps"])
        assert np.array_equal(res, G["residual"]) and np.array_equal(st, G["status"])
        pera = np.ascontiguousarray(_era(G["p_erased_bits"], n))
        pout, psw, pres, pst = ctx.decode(h, G["p_sym"], pera)
        assert np.array_equal(pout, G["p_out"]) and np.array_equal(psw, G["p_sweeps"]) and np.array_equal(pres, G["p_residual"])
        rs = ctx.rs_create(int(G["rs_n"]), int(G["rs_k"]))
        assert np.array_equal(ctx.rs_decode(rs, G["rs_idx"], G["rs_val"]), G["rs_msg"])
