"""Every knob of DESIGN.md's appendix, at every non-default value, set through ldpc_amd_configure (the LDPC_AMD_* environment
variables only give the initial values, read once by ldpc_amd_init: checked below with a second context).  A knob changes HOW a batch is decoded (kernel formulation, tiers, piece sizes, thread counts,
fall-backs) -- never a byte of the result: outputs, sweep counts, residual counts and status words must equal the default
run's, which in turn is checked against the oracle.  The batch mixes frames message passing completes, frames that need the
ML stage and rank-deficient ones, at S = 1 and as packets."""
import os

import numpy as np
import pytest

from ldpc_erasure_codes_amd import api, codes, synth

pytestmark = pytest.mark.gpu

KNOBS = [
    ("LDPC_AMD_APPLY", ["gather"]),
    ("LDPC_AMD_SCATTER_TIERS", ["1"]),
    ("LDPC_AMD_SCATTER_B", ["128", "64"]),
    ("LDPC_AMD_SCATTER_R", ["1", "4"]),
    ("LDPC_AMD_SCATTER_R2", ["2", "4"]),
    ("LDPC_AMD_SCATTER_T2B", ["128"]),
    ("LDPC_AMD_SCATTER_XL", ["0"]),
    ("LDPC_AMD_SCATTER_PAIRS", ["0"]),
    ("LDPC_AMD_SCATTER_T2P", ["1", "4"]),
    ("LDPC_AMD_SCATTER_LISTS", ["1"]),
    ("LDPC_AMD_SCATTER_NT", ["0"]),
    ("LDPC_AMD_SCATTER_XCD", ["0"]),
    ("LDPC_AMD_SCATTER_DYN", ["0", "2", "3", "4"]),
    ("LDPC_AMD_ENC_B", ["256"]),
    ("LDPC_AMD_ENC_LIST", ["1"]),
    ("LDPC_AMD_ENC_CLIST", ["0"]),
    ("LDPC_AMD_ENC_GROUP", ["0"]),
    ("LDPC_AMD_ENC_PERSIST", ["0"]),
    ("LDPC_AMD_PEEL_RELAX", ["0"]),
    ("LDPC_AMD_PEEL_GT", ["0", "1"]),
    ("LDPC_AMD_PEEL_WPB", ["1", "4", "16"]),
    ("LDPC_AMD_ML_SOLVE", ["0"]),
    ("LDPC_AMD_ML_SOLVE_B", ["64", "32", "16"]),
    ("LDPC_AMD_ML_ARENA_WORDS", ["20000"]),
    ("LDPC_AMD_ML_THREADS", ["256", "512", "768"]),
    ("LDPC_AMD_ML_PACK", ["1", "3", "4"]),
    ("LDPC_AMD_ML_PI", ["0", "2"]),                 # (=2, unverified: same bytes because this batch's symbols ARE codewords)
    ("LDPC_AMD_ML_OVERLAP", ["0", "1"]),
    ("LDPC_AMD_ML_OVERLAP_PRIO", ["1"]),
    ("LDPC_AMD_ML_PI_IMAX", ["0", "3", "17"]),
    ("LDPC_AMD_ML_PI_LDS", ["64", "96"]),
    ("LDPC_AMD_ML_PI_WAVES", ["1", "2"]),
    ("LDPC_AMD_ML_PI_ADAPTIVE", ["0"]),
    ("LDPC_AMD_RS", ["generic"]),
]


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _batch(code, S, F, seed):
    src = synth.source(seed, 0, F, code.k, S)
    # erasure rates from "a few sweeps" over "ML stage" to "rank deficient" for the (2040,1530) code
    pers = np.linspace(0.05, 0.262, F)
    era = np.concatenate([synth.erasures_uniform(seed + 1 + i, i, 1, code.n, float(p)) for i, p in enumerate(pers)])
    return (src if S > 1 else src[:, :, 0]), era


@pytest.mark.parametrize("S", [1, 256])
def test_every_knob_value_gives_the_default_bytes(ctx, oracle, code_a, S):
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    F = 48
    src, era = _batch(code_a, S, F, 900 + S)
    saved = {k: os.environ.pop(k) for k, _ in KNOBS if k in os.environ}
    for k, _ in KNOBS:
        ctx.configure(k, None)
    try:
        cw0 = ctx.encode(h, src)
        sym = cw0.copy()
        sym[era.astype(bool)] = 0x5A
        ref = ctx.decode(h, sym, era)
        out0, sw0, res0, st0 = ref
        assert {0, 1}.issubset(set(st0.tolist())) and st0.max() >= 2      # every kind of frame is in the batch
        oc = oracle.OracleCode(code_a)
        for f in (0, F // 2, F - 1):                                       # the default run against the oracle
            if S == 1:
                o = oc.decode_batch_s1(sym[f:f + 1], era[f:f + 1])
                assert np.array_equal(out0[f], o[0][0]) and sw0[f] == o[1][0] and res0[f] == o[2][0] and st0[f] == o[3][0]
            else:
                o, _, it, info, rc = oc.decode_packets(sym[f], era[f])
                assert np.array_equal(out0[f], o) and sw0[f] == it and res0[f] == info[0]
        for knob, values in KNOBS:
            for v in values:
                ctx.configure(knob, v)
                try:
                    cw = ctx.encode(h, src)
                    assert np.array_equal(cw, cw0), (knob, v, "encode")
                    out, sw, res, st = ctx.decode(h, sym, era)
                    assert np.array_equal(sw, sw0) and np.array_equal(res, res0) and np.array_equal(st, st0), (knob, v)
                    assert np.array_equal(out, out0), (knob, v)
                finally:
                    ctx.configure(knob, None)
    finally:
        os.environ.update(saved)


def test_environment_is_read_once_at_init_and_configure_rejects_nonsense(oracle, code_a):
    """LDPC_AMD_* variables are the INITIAL knob values of a context (read inside ldpc_amd_init, nowhere else): a context
    created under LDPC_AMD_APPLY=gather launches the gather kernel, changing the variable afterwards changes nothing, and an
    existing context is not affected by the environment at all.  Unknown keys and out-of-range values are refused."""
    h_src = synth.source(5, 0, 4, code_a.k, 64)
    era = synth.erasures_uniform(6, 0, 4, code_a.n, 0.10)
    os.environ["LDPC_AMD_APPLY"] = "gather"
    try:
        with api.Context(0) as c2:
            os.environ["LDPC_AMD_APPLY"] = "scatter"      # too late for c2
            h = c2.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
            cw = c2.encode(h, h_src)
            sym = cw.copy()
            sym[era.astype(bool)] = 0
            out = c2.decode(h, sym, era)[0]
            assert np.array_equal(out, cw)
            assert c2.last_plan()["packet_bytes_per_workgroup"] == 0          # gather form: no packet (scatter) kernel plan
            assert c2.profile_kernel_names()["apply"] == "ldpc_apply_kernel"
            c2.configure("APPLY", "scatter")                                   # ... the configure call is how it changes
            assert np.array_equal(c2.decode(h, sym, era)[0], cw)
            assert c2.last_plan()["packet_bytes_per_workgroup"] == 64
            assert c2.knobs() == ""                                            # everything back at its default ...
            c2.configure("ML_PI", "2"); c2.configure("RS", "generic")
            assert c2.knobs() == "ML_PI=2 RS=1"                                # ... and what is not is named (measurement scripts echo it)
            c2.configure("ML_PI", None); c2.configure("RS", None)
            for key, val in (("NO_SUCH_KNOB", "1"), ("SCATTER_B", "100"), ("ML_THREADS", "1000"), ("APPLY", "sideways")):
                with pytest.raises(api.LdpcAmdError):
                    c2.configure(key, val)
            assert np.array_equal(c2.decode(h, sym, era)[0], cw)              # a refused value leaves the knob as it was
    finally:
        os.environ.pop("LDPC_AMD_APPLY", None)


def test_a_rejected_environment_value_fails_init():
    """a value a knob does not take must not silently leave the default (an A/B script with a typo would measure the
    default under the wrong label): ldpc_amd_init refuses, like ldpc_amd_configure does, and names the variable."""
    for name, val in (("LDPC_AMD_ML_PI_LDS", "16"), ("LDPC_AMD_SCATTER_B", "banana"), ("LDPC_AMD_ML_ARENA_WORDS", str(1 << 40))):
        os.environ[name] = val
        try:
            with pytest.raises(api.LdpcAmdError, match=name):
                api.Context(0)
        finally:
            os.environ.pop(name, None)
    with api.Context(0) as c:      # and a clean environment still initialises
        c.selftest()
