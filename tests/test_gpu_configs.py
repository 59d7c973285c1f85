"""BASELINE.json configs 3-5 as parity-test cases (configs[1] is the bench; configs[0] is the binary CPU row,
tests/test_gpu_parity.py::test_binary_code_matches_binary_reference_decoder):

  cfg 3  n=2040,k=1530 hybrid-ML decode, bursty-channel erasures (ML stage forced on >= 10 % of the frames)
  cfg 4  n=4080,k=3060 GF(256) LDPC vs RS(255,223) side by side on the same erasure patterns, 65536 frames
  cfg 5  mixed n=4000,k=2000 + n=2040,k=1530 frame stream, bucketed by code and sharded

Small samples are compared bit for bit with the CPU oracle; the full sizes are checked through properties
(encode -> erase -> decode == codeword; RS decode == source) because the oracle would take minutes.
The (4080,3060) matrix is synthesised (tools/hgen.cpp) -- the reference names it but does not ship it.
"""
import numpy as np
import pytest

from ldpc_erasure_codes_amd import api, codes, sharding, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tctx():
    torch = pytest.importorskip("torch")
    c = api.Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


# ------------------------------------------------------------------------------------------ cfg 3
def test_cfg3_hybrid_ml_bursty_channel(tctx, oracle, code_a):
    ctx = tctx
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    oc = oracle.OracleCode(code_a)
    nframes = 256
    # the repository's own parameters (alpha <= 0.1, beta = 0.4) never reach the ML stage:
    # a heavier bad state is used, and the ML-trigger rate is asserted
    era = synth.erasures_bursty(31, 0, nframes, code_a.n, 0.13, 0.8, 10.0)
    keep = era.sum(axis=1) < code_a.m  # harness guard: decoder called only if num_erasures < n-k (...Sim.m:216)
    era = np.ascontiguousarray(era[keep])
    nf = era.shape[0]
    src = synth.source(32, 0, nf, code_a.k, 1)[:, :, 0]
    cw = ctx.encode(h, src)
    sym = cw.copy()
    sym[era.astype(bool)] = 0
    out, sw, res, st = ctx.decode(h, sym, era)
    o_out, o_sw, o_res, o_st = oc.decode_batch_s1(sym, era)
    assert np.array_equal(sw, o_sw) and np.array_equal(res, o_res) and np.array_equal(st, o_st)
    assert np.array_equal(out, o_out)
    ml_rate = float((res > 0).mean())
    assert ml_rate >= 0.10, f"ML stage triggered on only {ml_rate:.1%} of the frames"
    ok = np.isin(st, (0, 1))
    assert np.array_equal(out[ok], cw[ok])
    # same patterns as 64-byte packets: lanes share the schedule and the elimination
    S = 64
    sub = np.nonzero(res > 0)[0][:6].tolist() + np.nonzero(res == 0)[0][:2].tolist()
    psrc = synth.source(33, 0, len(sub), code_a.k, S)
    pcw = ctx.encode(h, psrc)
    pera = np.ascontiguousarray(era[sub])
    psym = pcw.copy()
    psym[pera.astype(bool)] = 0x77
    pout, psw, pres, pst = ctx.decode(h, psym, pera)
    for i, f in enumerate(sub):
        o, _, it, info, rc = oc.decode_packets(psym[i], pera[i])
        assert np.array_equal(pout[i], o) and psw[i] == it == sw[f] and pres[i] == res[f]


# ------------------------------------------------------------------------------------------ cfg 4
def rs_side_by_side(ctx, rs, era, rs_n, rs_k, S=1, seed=41):
    """16 RS(255,223) blocks per 4080-symbol frame on the same erasure pattern (block b = symbols 255b..255b+254,
    Matlab/ErasureCodes_NonBinaryLDPCSim.m:210-214).  Returns (decodable mask [F,16], ok mask [F,16])."""
    F, n = era.shape
    nb = n // rs_n
    blocks = era[:, : nb * rs_n].reshape(F * nb, rs_n)
    received = blocks == 0
    can = received.sum(axis=1) >= rs_k  # RS decode attempted only with >= k received (ReedSolomonErasureCodes.m:80)
    idx_all = np.argsort(~received, axis=1, kind="stable")[:, :rs_k].astype(np.uint16)  # first k received, ascending
    sel = np.nonzero(can)[0]
    rng = np.random.default_rng(seed)
    src = rng.integers(0, 256, size=(sel.size, rs_k) if S == 1 else (sel.size, rs_k, S)).astype(np.uint8)
    cw = ctx.rs_encode(rs, rs_n, rs_k, src)
    idx = np.ascontiguousarray(idx_all[sel])
    val = np.take_along_axis(cw, idx.astype(np.int64) if S == 1 else idx.astype(np.int64)[:, :, None], axis=1)
    msg = ctx.rs_decode(rs, idx, np.ascontiguousarray(val))
    ok = np.zeros(F * nb, dtype=bool)
    ok[sel] = (msg == src).reshape(sel.size, -1).all(axis=1)
    return can.reshape(F, nb), ok.reshape(F, nb), (idx, val, msg, src)


def test_cfg4_code_c_vs_rs_small_sample_against_oracle(tctx, oracle):
    ctx = tctx
    code = codes.load_builtin(3)
    assert (code.n, code.k) == (4080, 3060)
    h = ctx.load_builtin_code(3, codes.DEFAULT_COEF_SEED[3])
    oc = oracle.OracleCode(code)
    nframes = 24
    era = np.concatenate([synth.erasures_uniform(42, 0, 16, code.n, 0.10), synth.erasures_uniform(43, 0, 8, code.n, 0.21)])
    src = synth.source(44, 0, nframes, code.k, 1)[:, :, 0]
    cw = ctx.encode(h, src)
    assert np.array_equal(cw[0], oc.encode(src[0]))
    sym = cw.copy()
    sym[era.astype(bool)] = 0
    out, sw, res, st = ctx.decode(h, sym, era)
    o_out, o_sw, o_res, o_st = oc.decode_batch_s1(sym, era)
    assert np.array_equal(out, o_out) and np.array_equal(sw, o_sw) and np.array_equal(st, o_st)
    rs = ctx.rs_create(255, 223)
    G = ctx.rs_generator(rs, 255, 223)
    can, ok, (idx, val, msg, rsrc) = rs_side_by_side(ctx, rs, era, 255, 223)
    assert ok[can].all()
    for b in range(0, idx.shape[0], 37):
        o, rc = oracle.rs_decode(G, idx[b], val[b])
        assert rc == 0 and np.array_equal(o, msg[b])
    # at 10 % the rate-0.875 RS blocks fail often (> 32 erasures of 255), the rate-0.75 LDPC code does not
    assert (st[:16] == 0).all()


def test_cfg4_full_size_65536_frames(tctx):
    torch = pytest.importorskip("torch")
    ctx = tctx
    code = codes.load_builtin(3)
    h = ctx.load_builtin_code(3, codes.DEFAULT_COEF_SEED[3])
    F = 65536
    dev = torch.device("cuda", 0)
    src = torch.empty((F, code.k), dtype=torch.uint8, device=dev)
    ctx.synth_source(45, 0, F, code.k, 1, src)
    cw = ctx.encode(h, src)
    era = torch.empty((F, code.n), dtype=torch.uint8, device=dev)
    ctx.synth_erasures_uniform(46, 0, F, code.n, 0.10, era)
    sym = cw.clone()
    sym[era.bool()] = 0xEE
    out, sw, res, st = ctx.decode(h, sym, era)
    ctx.synchronize()
    assert torch.equal(out, cw) and int(st.max()) <= 1
    # RS side: all 16 x 65536 blocks of the same patterns, checked on the device
    rs = ctx.rs_create(255, 223)
    blocks = era.reshape(F * 16, 255)
    received = blocks == 0
    can = received.sum(dim=1) >= 223
    order = torch.argsort((~received).to(torch.uint8), dim=1, stable=True)[:, :223]
    sel = torch.nonzero(can).flatten()
    B = int(sel.numel())
    assert 0.3 * F * 16 < B < F * 16  # many, not all, blocks are decodable at 10 %
    rsrc = torch.empty((B, 223), dtype=torch.uint8, device=dev)
    ctx.synth_source(47, 0, B, 223, 1, rsrc)
    rcw = ctx.rs_encode(rs, 255, 223, rsrc)
    idx = order[sel].to(torch.int16).contiguous()
    val = torch.gather(rcw, 1, order[sel]).contiguous()
    msg = ctx.rs_decode(rs, idx, val)
    ctx.synchronize()
    assert torch.equal(msg, rsrc)


# ------------------------------------------------------------------------------------------ cfg 5
def test_cfg5_mixed_stream_bucketed_and_sharded(tctx, oracle, code_a, code_b):
    ctx = tctx
    ha = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    hb = ctx.load_builtin_code(2, codes.DEFAULT_COEF_SEED[2])
    handles = {1: (ha, code_a), 2: (hb, code_b)}
    total = 64
    ids = np.array([2, 1] * (total // 2))  # interleaved 1:1 stream of code B and code A frames
    oracles = {1: oracle.OracleCode(code_a), 2: oracle.OracleCode(code_b)}
    per = {1: 0.10, 2: 0.30}
    results = {}
    world = 8
    for rank in range(world):  # every simulated rank decodes its share with the same context
        for cid, gidx in sharding.shard_mixed(ids, rank, world).items():
            if gidx.size == 0:
                continue
            hnd, code = handles[cid]
            era = np.concatenate([synth.erasures_uniform(50 + cid, int(g), 1, code.n, per[cid]) for g in gidx])
            src = np.concatenate([synth.source(60 + cid, int(g), 1, code.k, 1)[:, :, 0] for g in gidx])
            cw = ctx.encode(hnd, src)
            sym = cw.copy()
            sym[era.astype(bool)] = 0
            out, sw, res, st = ctx.decode(hnd, sym, era)
            for i, g in enumerate(gidx):
                results[int(g)] = (cid, out[i], sw[i], st[i], cw[i], sym[i], era[i])
    assert sorted(results) == list(range(total))
    for g in range(0, total, 5):
        cid, out, sw, st, cw, sym, era = results[g]
        assert cid == ids[g]
        o, osw, ores, ost = oracles[cid].decode_batch_s1(sym[None], era[None])
        assert np.array_equal(o[0], out) and osw[0] == sw and ost[0] == st
    assert all(np.array_equal(r[1], r[4]) for r in results.values() if r[3] in (0, 1))


# ------------------------------------------------------------------------------------------ full-size batches
class _Args:
    pass


@pytest.fixture(scope="module")
def bgpu():
    """bench.py's own device-side generators (the batches the driver-run bench line is measured on)."""
    pytest.importorskip("torch")
    import bench
    g = bench.Gpu(_Args(), 0, 1, 0)
    yield g
    g.close()


def _lane(t, lane):
    return t[:, :, lane].contiguous()


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
def test_full_4096_frame_packet_batches(bgpu, oracle, code_a, cfg):
    """BASELINE cfg 2 and cfg 3 at their full size (4096 frames drawn, S = 1024) -- the batches bench.py times -- through
    size-independent properties: every frame the decoder reports solved equals its codeword; every byte lane of the packet
    decode equals the S = 1 (Matlab-model) decode of that lane, rank-deficient frames included ("each byte lane
    of an S > 1 decode must equal an S = 1 decode of that lane with the same pattern"); the packet ML stage's two
    implementations (solve schedules in LDS slices / inside the ML kernel) agree byte for byte; a few frames of every status
    are compared with the oracle."""
    import os
    import torch
    g = bgpu
    h, n, k = g.code(1)
    cw, sym, era, keep = g.make_batch(cfg, 1, 1024, frame0=0, nframes=4096)
    F = cw.shape[0]
    out, sw, res, st = g.ctx.decode(h, sym, era)
    torch.cuda.synchronize()
    ok = st <= 1
    assert torch.equal(out[ok], cw[ok])
    if cfg == "cfg2":
        assert F == 4096 and int(st.max()) == 0
    else:
        assert F < 4096 and float((res > 0).float().mean()) >= 0.10 and int((st == 2).sum()) > 0   # ML runs, some systems are rank deficient
        g.ctx.configure("LDPC_AMD_ML_SOLVE", 0)
        try:
            out2, sw2, res2, st2 = g.ctx.decode(h, sym, era)
            torch.cuda.synchronize()
        finally:
            g.ctx.configure("LDPC_AMD_ML_SOLVE", None)
        assert torch.equal(out2, out) and torch.equal(st2, st) and torch.equal(sw2, sw)
        # a small schedule arena: the first frames emit their solve schedules, the rest are solved inside the ML kernel --
        # both kinds in one batch, same bytes
        g.ctx.configure("LDPC_AMD_ML_ARENA_WORDS", 1 << 21)
        try:
            out3, sw3, res3, st3 = g.ctx.decode(h, sym, era)
            torch.cuda.synchronize()
        finally:
            g.ctx.configure("LDPC_AMD_ML_ARENA_WORDS", None)
        assert torch.equal(out3, out) and torch.equal(st3, st)
        del out2, out3
    for lane in (0, 777):
        o1, sw1, res1, st1 = g.ctx.decode(h, _lane(sym, lane), era)
        torch.cuda.synchronize()
        assert torch.equal(sw1, sw) and torch.equal(res1, res) and torch.equal(st1, st)
        assert torch.equal(o1, _lane(out, lane))
    oc = oracle.OracleCode(code_a)
    picks = [0]
    for code in (1, 2):
        hit = (st == code).nonzero().flatten()
        if hit.numel():
            picks.append(int(hit[0]))
    for f in picks:
        o, _, it, info, _ = oc.decode_packets(sym[f].cpu().numpy(), era[f].cpu().numpy())
        assert np.array_equal(o, out[f].cpu().numpy()) and it == int(sw[f]) and info[0] == int(res[f])


def test_long_batch_crosses_the_internal_chunks(bgpu):
    """Batches above 16 384 frames are processed in chunks that share the per-call workspaces (schedules, ML lists, the arena
    of the ML solve schedules): a 40 000-frame bursty batch (S = 64, ML stage on a quarter of the frames) must decode exactly
    like its two halves decoded separately, and every solved frame must equal its codeword."""
    import torch
    g = bgpu
    h, n, k = g.code(1)
    cw, sym, era, keep = g.make_batch("cfg3", 1, 64, frame0=0, nframes=40000)
    F = cw.shape[0]
    assert F > 2 * 16384
    out, sw, res, st = g.ctx.decode(h, sym, era)
    torch.cuda.synchronize()
    assert float((res > 0).float().mean()) > 0.1
    ok = st <= 1
    assert torch.equal(out[ok], cw[ok])
    half = F // 2
    for lo, hi in ((0, half), (half, F)):
        o2, sw2, res2, st2 = g.ctx.decode(h, sym[lo:hi].contiguous(), era[lo:hi].contiguous())
        torch.cuda.synchronize()
        assert torch.equal(o2, out[lo:hi]) and torch.equal(st2, st[lo:hi]) and torch.equal(sw2, sw[lo:hi]) and torch.equal(res2, res[lo:hi])
