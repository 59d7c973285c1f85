"""Round-3 GPU parity tests: RS erasure decoding in packet mode at a few hundred blocks against the oracle, the whole
65 536-frame mixed stream of BASELINE cfg 5, (4080,3060) in packet mode at batch size, the malformed-block signal of the
device-pointer RS path, and the configure call.  All through the C ABI (ctypes binding), checker = oracle/ (CPU restatement
of Matlab/My_RS_Decode_Optimize_With_GFTables.m and Matlab/My_LDPC_HybridML_NonBinary_Erasure_Decoder.m)."""
import numpy as np
import pytest

from ldpc_erasure_codes_amd import api, codes, sharding, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _rs_case(rng, n, k, B, S):
    """B blocks with every kind of pattern: nothing missing, one missing, n-k missing (all repair symbols needed), only
    repair symbols at the end, random counts."""
    idx = np.zeros((B, k), dtype=np.uint16)
    for b in range(B):
        special = [0, 1, n - k, (n - k) // 2, n - k - 1]
        nerase = special[b] if b < len(special) else int(rng.integers(0, n - k + 1))
        if b == 5:      # the first n-k source symbols missing: the block starts with a run of repair work
            keep = np.arange(n - k, n)
        elif b == 6:    # the last source symbols missing
            keep = np.concatenate([np.arange(0, k - (n - k)), np.arange(k, n)])
        else:
            keep = np.sort(rng.permutation(n)[: n - nerase])
        idx[b] = keep[:k]
    return idx


@pytest.mark.parametrize("n,k,S,B", [(255, 223, 1024, 300), (255, 223, 256, 200), (255, 223, 512, 120), (255, 223, 2048, 40),
                                     (255, 223, 64, 100), (7, 5, 256, 64), (255, 192, 1024, 24), (250, 125, 256, 12)])
def test_rs_decode_packets_hundreds_of_blocks(ctx, oracle, n, k, S, B):
    """Packet-mode RS decode (streaming kernel for n-k <= 32 and S a multiple of 256, generic kernel otherwise): every block
    equals its source, and byte lanes of sampled blocks equal the oracle's S = 1 decode of that lane -- also on NON-codeword
    input (any k received values of an MDS code are consistent, so the reference's elimination has one answer)."""
    rs = ctx.rs_create(n, k)
    G = ctx.rs_generator(rs, n, k)
    rng = np.random.default_rng(n * 1000 + S)
    src = rng.integers(0, 256, size=(B, k, S)).astype(np.uint8)
    cw = ctx.rs_encode(rs, n, k, src)
    idx = _rs_case(rng, n, k, B, S)
    val = np.stack([cw[b, idx[b].astype(np.int64)] for b in range(B)])
    msg = ctx.rs_decode(rs, idx, val)
    assert np.array_equal(msg, src)
    assert ctx.rs_bad_blocks() == 0
    for vw in (2, 4):                      # the wider-lane instantiations of the streaming kernel (A/B knob)
        ctx.configure("RS_VW", vw)
        try:
            assert np.array_equal(ctx.rs_decode(rs, idx, val), src), vw
        finally:
            ctx.configure("RS_VW", None)
    # arbitrary received values (not a codeword of anything in particular)
    junk = rng.integers(0, 256, size=val.shape).astype(np.uint8)
    jmsg = ctx.rs_decode(rs, idx, junk)
    for b in list(range(0, min(B, 8))) + list(range(8, B, max(1, B // 12))):
        for lane in (0, S // 3, S - 1):
            o, rc = oracle.rs_decode(G, idx[b], np.ascontiguousarray(val[b, :, lane]))
            assert rc == 0 and np.array_equal(msg[b, :, lane], o), (b, lane)
            o, rc = oracle.rs_decode(G, idx[b], np.ascontiguousarray(junk[b, :, lane]))
            assert np.array_equal(jmsg[b, :, lane], o), (b, lane)


def test_rs_packets_device_pointers_signal_malformed_blocks(ctx):
    """Device-pointer path: a block whose positions are not strictly ascending (or >= n) decodes to zeros AND is counted, so a
    caller can tell it from a valid all-zero message; the other blocks are untouched."""
    torch = pytest.importorskip("torch")
    n, k = 255, 223
    rs = ctx.rs_create(n, k)
    dev = torch.device("cuda", 0)
    for S in (1, 1024, 64):
        B = 9
        rng = np.random.default_rng(S)
        shape = (B, k) if S == 1 else (B, k, S)
        src = rng.integers(0, 256, size=shape).astype(np.uint8)
        cw = ctx.rs_encode(rs, n, k, src)
        idx = np.tile(np.concatenate([np.arange(0, k - 10), np.arange(k, k + 10)]).astype(np.uint16), (B, 1))
        val = cw[:, idx[0].astype(np.int64)]
        idx[2, 5] = idx[2, 4]            # not strictly ascending
        idx[7, k - 1] = n                # position out of range
        t_idx = torch.from_numpy(idx.view(np.int16)).to(dev)
        t_val = torch.from_numpy(np.ascontiguousarray(val)).to(dev)
        msg = ctx.rs_decode(rs, t_idx, t_val)
        ctx.synchronize()
        assert ctx.rs_bad_blocks() == 2
        m = msg.cpu().numpy()
        good = [b for b in range(B) if b not in (2, 7)]
        assert np.array_equal(m[good], src[good]) and not m[2].any() and not m[7].any()
        # a clean call resets the count
        idx[2], idx[7] = idx[0], idx[0]
        msg = ctx.rs_decode(rs, torch.from_numpy(idx.view(np.int16)).to(dev), t_val)
        assert ctx.rs_bad_blocks() == 0
        assert np.array_equal(msg.cpu().numpy(), src)


def test_cfg5_whole_65536_frame_mixed_stream():
    """BASELINE configs[4] at full size on one GPU (the N = 1 anchor of the strong-scaling job): 65 536 frames, codes
    (4000,2000) and (2040,1530) interleaved 1:1, bucketed by code; encode -> erase -> decode == codeword for both buckets,
    and the final gather returns every frame's status words once, in stream order."""
    torch = pytest.importorskip("torch")
    import bench

    class A:
        steps, warmup = 3, 1
    g = bench.Gpu(A(), 0, 1, 0)
    try:
        total = bench.WORKLOADS["cfg5"]["frames"]
        assert total == 65536
        ids = sharding.mixed_stream_ids(total)
        mine = sharding.shard_mixed(ids, 0, 1)
        shard = {}
        for ci, gidx in mine.items():
            h, n, k = g.code(ci)
            cw, sym, era, keep = g.make_batch("cfg5", ci, 1, frame_ids=gidx)
            assert bool(keep.all()) and cw.shape == (total // 2, n)
            out, sw, res, st = g.ctx.decode(h, sym, era)
            g.ctx.synchronize()
            assert torch.equal(out, cw) and int(st.max()) <= 1
            assert int(sw.min()) >= 1 and int(sw.max()) <= 10
            shard[ci] = {"gidx": gidx, "out": out, "words": torch.stack([sw, res, st])}
        full = sharding.gather_mixed(ids, shard, 1, "status")
        assert sorted(full) == [1, 2]
        for ci, v in full.items():
            assert v["words"].shape == (3, total // 2) and np.array_equal(v["gidx"], np.nonzero(ids == ci)[0])
        # the 1:1 interleave: code B on even stream positions, code A on odd ones
        assert (full[2]["gidx"] % 2 == 0).all() and (full[1]["gidx"] % 2 == 1).all()
    finally:
        g.close()


def test_code_c_packets_at_batch_size(oracle):
    """(4080,3060) on the packet path: 512 frames x 1 KB symbols, uniform 10 % -- round trip for the batch,
    byte lanes of sampled frames against the oracle's S = 1 decode, and the launch plan the bench line reports."""
    torch = pytest.importorskip("torch")
    if not codes.have_builtin(3):
        pytest.skip("(4080,3060) not built in")
    code = codes.load_builtin(3)
    oc = oracle.OracleCode(code)
    with api.Context(0) as ctx:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)   # torch ops below and the library's kernels: one stream
        h = ctx.load_builtin_code(3, codes.DEFAULT_COEF_SEED[3])
        dev = torch.device("cuda", 0)
        F, S = 512, 1024
        src = torch.empty((F, code.k, S), dtype=torch.uint8, device=dev)
        ctx.synth_source(71, 0, F, code.k, S, src)
        cw = ctx.encode(h, src)
        era = torch.empty((F, code.n), dtype=torch.uint8, device=dev)
        ctx.synth_erasures_uniform(72, 0, F, code.n, 0.10, era)
        era[5] = 0                                   # a frame without erasures
        era[6, ::7] = 1                              # ... and heavier ones (14 % and 20 %: deeper schedules, tier 2)
        era[7, ::5] = 1
        sym = cw.clone()
        sym[era.bool()] = 0x3C
        out, sw, res, st = ctx.decode(h, sym, era)
        ctx.synchronize()
        ok = st <= 1
        assert torch.equal(out[ok], cw[ok]) and int(ok.sum()) >= F - 1
        plan = ctx.last_plan()
        assert plan["packet_bytes_per_workgroup"] in (64, 128, 256)
        e_np, s_np, o_np = era.cpu().numpy(), sym.cpu().numpy(), out.cpu().numpy()
        for f in (0, 5, 6, 7, F - 1):
            for lane in (0, 517, S - 1):
                o, osw, ores, ost = oc.decode_batch_s1(np.ascontiguousarray(s_np[f, :, lane])[None], e_np[f][None])
                assert np.array_equal(o[0], o_np[f, :, lane]) and osw[0] == int(sw[f]) and ost[0] == int(st[f]), (f, lane)


def test_host_pipeline_for_encode_and_rs_matches_single_shot(ctx, code_a):
    """Host pointers above the pipeline threshold (192 MB): LDPC encode, RS encode and RS decode go through the same chunked
    upload / compute / download pipeline as the decoder -- same bytes as the single-shot path
    (HOST_PIPELINE = 0), ragged last chunk included."""
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    S = 1024
    src = synth.source(811, 0, 110, code_a.k, S)                 # 110 x 2.09 MB = 230 MB of codewords: 3 chunks (45/45/20)
    rs = ctx.rs_create(255, 223)
    rng = np.random.default_rng(812)
    rsrc = rng.integers(0, 256, size=(900, 223, S)).astype(np.uint8)   # 205 MB in, 235 MB out of the encoder
    idx = np.stack([np.sort(rng.permutation(255)[:255 - int(rng.integers(0, 33))])[:223] for _ in range(900)]).astype(np.uint16)
    ctx.configure("HOST_PIPELINE", 0)
    try:
        cw_ref = ctx.encode(h, src)
        rcw_ref = ctx.rs_encode(rs, 255, 223, rsrc)
        val = np.stack([rcw_ref[b, idx[b].astype(np.int64)] for b in range(900)])
        msg_ref = ctx.rs_decode(rs, idx, val)
    finally:
        ctx.configure("HOST_PIPELINE", None)
    assert np.array_equal(msg_ref, rsrc)
    assert np.array_equal(ctx.encode(h, src), cw_ref)
    assert np.array_equal(ctx.rs_encode(rs, 255, 223, rsrc), rcw_ref)
    assert np.array_equal(ctx.rs_decode(rs, idx, val), rsrc)
    assert ctx.rs_bad_blocks() == 0


def test_small_host_calls_take_the_packed_path(ctx, oracle, code_a):
    """A single Matlab-style frame (and anything up to 1 MB) goes up as ONE pinned transfer and comes back as one: same
    results as the oracle, optional outputs still optional, and the next larger size class (separate transfers) agrees."""
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    oc = oracle.OracleCode(code_a)
    for F, per in ((1, 0.10), (1, 0.215), (7, 0.19), (100, 0.12), (300, 0.10)):      # 300 frames x 2040 B: above the 1 MB class
        src = synth.source(820 + F, 0, F, code_a.k, 1)[:, :, 0]
        cw = ctx.encode(h, src)
        era = synth.erasures_uniform(830 + F, 0, F, code_a.n, per)
        sym = cw.copy()
        sym[era.astype(bool)] = 0x11
        out, sw, res, st = ctx.decode(h, sym, era)
        o_out, o_sw, o_res, o_st = oc.decode_batch_s1(sym, era)
        assert np.array_equal(out, o_out) and np.array_equal(sw, o_sw) and np.array_equal(res, o_res) and np.array_equal(st, o_st)
        L = ctx._L
        out2 = np.zeros_like(out)
        assert L.ldpc_amd_decode_batch(ctx._h, h, 1, F, sym.ctypes.data, era.ctypes.data, 10, 1, out2.ctypes.data, None, None, None, 0) == 0
        assert np.array_equal(out2, out)


def test_rs_packets_fuzz_over_code_shapes(ctx, oracle):
    """Seeded fuzz of the RS packet path over (n, k, S): every n - k <= 32 shape takes the streaming kernel (S a multiple of 256),
    others the generic one; random numbers of missing symbols per block.  Every block against its source, sampled byte lanes
    against the oracle (Matlab/My_RS_Decode_Optimize_With_GFTables.m restated)."""
    rng = np.random.default_rng(20261004)
    shapes = [(255, 223), (255, 239), (200, 180), (64, 32), (40, 36), (33, 1), (17, 16), (255, 224), (100, 60), (255, 128)]
    for (n, k) in shapes:
        rs = ctx.rs_create(n, k)
        G = ctx.rs_generator(rs, n, k)
        for S in (256, 768, 1280):
            B = 24
            src = rng.integers(0, 256, size=(B, k, S)).astype(np.uint8)
            cw = ctx.rs_encode(rs, n, k, src)
            idx = np.stack([np.sort(rng.permutation(n)[: n - int(rng.integers(0, n - k + 1))])[:k] for _ in range(B)]).astype(np.uint16)
            val = np.stack([cw[b, idx[b].astype(np.int64)] for b in range(B)])
            msg = ctx.rs_decode(rs, idx, val)
            assert np.array_equal(msg, src), (n, k, S)
            for b in (0, B // 2, B - 1):
                lane = int(rng.integers(0, S))
                o, rc = oracle.rs_decode(G, idx[b], np.ascontiguousarray(val[b, :, lane]))
                assert rc == 0 and np.array_equal(msg[b, :, lane], o), (n, k, S, b, lane)


def test_repeated_decodes_of_the_bench_batches_are_identical():
    """The accumulation ORDER of a decode is not fixed (LDS atomics, work items handed out first come first served, two ML
    systems per CU, the larger matrices in the L2-backed scratch); its RESULT is: the 4096-frame cfg 3 batch (hybrid ML,
    rank-deficient frames included) and the cfg 2 batch, decoded repeatedly, give the same bytes and status words every time
    (tools/stress_repeat.py; the first run is also held against the codewords)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_repeat.py"), "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("identical") == 5 and "NOT identical" not in r.stdout
