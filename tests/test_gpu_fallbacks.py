"""Shapes that leave the fast paths: heavy columns (no scatter kernel -> gather kernel), systems too large for the
LDS-resident ML matrix (global scratch), tier-2 scatter frames, S larger than one wavefront pass."""
import numpy as np
import pytest

from ldpc_erasure_codes_amd import api, codes, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def dense_code(n, k, rowdeg, seed):
    rng = np.random.default_rng(seed)
    m = n - k
    H = np.zeros((m, n), dtype=np.uint8)
    for i in range(m):
        c = rng.choice(k + i, size=min(rowdeg - 1, k + i), replace=False)
        H[i, c] = rng.integers(1, 256, size=c.size)
        H[i, k + i] = rng.integers(1, 256)
    return codes.from_dense(H, k)


def check_packets(ctx, oracle, code, h, S, pers, seed):
    oc = oracle.OracleCode(code)
    F = len(pers)
    src = synth.source(seed, 0, F, code.k, S)
    cw = ctx.encode(h, src)
    assert np.array_equal(cw[0], oc.encode(src[0]))
    era = np.concatenate([synth.erasures_uniform(seed + 1 + i, i, 1, code.n, p) for i, p in enumerate(pers)])
    sym = cw.copy()
    sym[era.astype(bool)] = 0x99
    out, sw, res, st = ctx.decode(h, sym, era)
    for f in range(F):
        o, _, it, info, rc = oc.decode_packets(sym[f], era[f])
        assert sw[f] == it and res[f] == info[0], f
        assert np.array_equal(out[f], o), (f, st[f])
    return st


def test_heavy_columns_use_the_gather_kernel(ctx, oracle):
    code = dense_code(48, 24, 22, 5)  # column degrees > 16: the padded per-symbol lists are not built
    assert np.bincount(code.cols, minlength=code.n).max() > 16
    h = ctx.register_code(code)
    st = check_packets(ctx, oracle, code, h, 32, [0.0, 0.1, 0.2, 0.3, 0.4, 0.45], 300)
    assert len(st) == 6


def test_large_residual_systems_use_the_global_ml_scratch(ctx, oracle, code_b):
    # (4000,2000): m = 2000, residuals of several hundred symbols do not fit the LDS matrix
    h = ctx.load_builtin_code(2, codes.DEFAULT_COEF_SEED[2])
    st = check_packets(ctx, oracle, code_b, h, 16, [0.45, 0.47, 0.48], 310)
    assert (st >= 1).any()
    oc = oracle.OracleCode(code_b)
    src = synth.source(320, 0, 4, code_b.k, 1)[:, :, 0]
    cw = ctx.encode(h, src)
    era = np.concatenate([synth.erasures_uniform(321 + i, i, 1, code_b.n, p) for i, p in enumerate((0.46, 0.47, 0.48, 0.485))])
    sym = cw.copy()
    sym[era.astype(bool)] = 0
    out, sw, res, st = ctx.decode(h, sym, era)
    o_out, o_sw, o_res, o_st = oc.decode_batch_s1(sym, era)
    assert np.array_equal(out, o_out) and np.array_equal(st, o_st) and np.array_equal(res, o_res)
    assert res.max() > 300


def test_tier2_frames_and_wide_packets(ctx, oracle, code_a):
    # > 267 solved symbols in a frame -> second scatter tier; S = 4096 -> 16 slices per row
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    st = check_packets(ctx, oracle, code_a, h, 1024, [0.16, 0.17, 0.10], 330)
    assert (st == 0).all()
    check_packets(ctx, oracle, code_a, h, 4096, [0.12, 0.2], 340)


def test_device_bursty_channel_equals_sequential_chain(ctx, oracle):
    """Parallel-scan Gilbert-Elliott generator == the oracle's literal sequential chain
    (Matlab/Bursty_Error_Channel_Model_Generator.m:12-47), incl. a start in the middle of the stream."""
    torch = pytest.importorskip("torch")
    n = 2040
    for frame0, nframes, alpha, beta in ((0, 40, 0.1, 0.4), (7, 33, 0.13, 0.8), (0, 1, 0.0, 1.0)):
        d = torch.empty((nframes, n), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.synth_erasures_bursty(31, frame0, nframes, n, alpha, beta, 10.0, d)
        ctx.synchronize()
        want = oracle.synth_erasures_bursty(31, frame0, nframes, n, alpha, beta, 10.0)
        assert np.array_equal(d.cpu().numpy(), want)
        assert np.array_equal(want, synth.erasures_bursty(31, frame0, nframes, n, alpha, beta, 10.0))


def test_ml_solve_kernel_geometries(ctx, oracle, code_b):
    """The packet ML stage's solve kernel picks its slice width from the number of checks: 128-byte slices for the (2040,1530)
    code, 32-byte slices for the (4080,3060) code, 16-byte slices for the (4000,2000) code.  Each geometry against the oracle,
    with residual systems that exist (status 1 or 2), at a packet size that gives several slices per row."""
    hb = ctx.load_builtin_code(2, codes.DEFAULT_COEF_SEED[2])
    st = check_packets(ctx, oracle, code_b, hb, 64, [0.46, 0.475, 0.30], 410)
    assert (st >= 1).any()
    if codes.have_builtin(3):
        code_c = codes.load_builtin(3)
        hc = ctx.load_builtin_code(3, codes.DEFAULT_COEF_SEED[3])
        st = check_packets(ctx, oracle, code_c, hc, 64, [0.215, 0.225, 0.23, 0.10], 420)
        assert (st >= 1).any()
        st = check_packets(ctx, oracle, code_c, hc, 1024, [0.22], 430)
