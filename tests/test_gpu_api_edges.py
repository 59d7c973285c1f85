"""GPU tests of the boundary itself: error behaviour (int codes + last_error, no exceptions across the ABI), empty and
ragged batches, properties the domain offers at sizes the oracle would not finish (linearity, idempotence), and the
C++ host harness that mirrors OpenCL/host/src/main.cpp."""
import os
import subprocess

import numpy as np
import pytest

from ldpc_erasure_codes_amd import api, codes, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def test_error_codes_and_messages(ctx, code_a):
    h = ctx.load_builtin_code(1, 2040)
    L = api.load_library()
    sym = np.zeros((1, code_a.n, 24), dtype=np.uint8)  # S = 24 is neither 1 nor a multiple of 16
    era = np.zeros((1, code_a.n), dtype=np.uint8)
    out = np.zeros_like(sym)
    rc = L.ldpc_amd_decode_batch(ctx._h, h, 24, 1, sym.ctypes.data, era.ctypes.data, 10, 1, out.ctypes.data, None, None, None, 0)
    assert rc == -5 and b"multiple of 16" in L.ldpc_amd_last_error(ctx._h)
    rc = L.ldpc_amd_decode_batch(ctx._h, 999, 1, 1, sym.ctypes.data, era.ctypes.data, 10, 1, out.ctypes.data, None, None, None, 0)
    assert rc == -4
    rc = L.ldpc_amd_decode_batch(ctx._h, h, 1, 1, None, era.ctypes.data, 10, 1, out.ctypes.data, None, None, None, 0)
    assert rc == -1
    rc = L.ldpc_amd_decode_batch(ctx._h, h, 1, 1, sym.ctypes.data, era.ctypes.data, 0, 1, out.ctypes.data, None, None, None, 0)
    assert rc == -1 and b"max_sweeps" in L.ldpc_amd_last_error(ctx._h)
    # nframes = 0 is a no-op, not an error
    assert L.ldpc_amd_decode_batch(ctx._h, h, 1, 0, None, None, 10, 1, None, None, None, None, 0) == 0
    with pytest.raises(api.LdpcAmdError):
        ctx.load_builtin_code(17, 1)
    with pytest.raises(api.LdpcAmdError):
        ctx.rs_create(300, 10)
    # malformed custom codes are rejected with a message
    bad = codes.from_dense(np.array([[1, 1, 0], [0, 1, 1]], dtype=np.uint8), 1)
    bad.cols = bad.cols[::-1].copy()
    with pytest.raises(api.LdpcAmdError):
        ctx.register_code(bad)


def test_status_pointers_are_optional_and_batches_can_be_ragged(ctx, oracle, code_a):
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    oc = oracle.OracleCode(code_a)
    L = api.load_library()
    for nframes in (1, 3, 5, 67):  # not multiples of the frames-per-workgroup
        src = synth.source(70, 0, nframes, code_a.k, 1)[:, :, 0]
        cw = ctx.encode(h, src)
        era = synth.erasures_uniform(71, 0, nframes, code_a.n, 0.15)
        sym = cw.copy()
        sym[era.astype(bool)] = 1
        out = np.zeros_like(sym)
        rc = L.ldpc_amd_decode_batch(ctx._h, h, 1, nframes, sym.ctypes.data, era.ctypes.data, 10, 1, out.ctypes.data, None, None, None, 0)
        assert rc == 0 and np.array_equal(out, cw)
    assert np.array_equal(cw[0], oc.encode(src[0]))


def test_linearity_and_idempotence_at_full_batch_size(code_a):
    """Decoding is GF(256)-linear in the received values for a fixed erasure pattern, and decoding a decoded word
    again (nothing erased) returns it unchanged -- checked on 4096 frames of 64-byte packets."""
    torch = pytest.importorskip("torch")
    ctx = api.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    F, S, n, k = 4096, 64, code_a.n, code_a.k
    dev = torch.device("cuda", 0)
    a = torch.empty((F, k, S), dtype=torch.uint8, device=dev)
    b = torch.empty((F, k, S), dtype=torch.uint8, device=dev)
    ctx.synth_source(81, 0, F, k, S, a)
    ctx.synth_source(82, 0, F, k, S, b)
    ca, cb = ctx.encode(h, a), ctx.encode(h, b)
    cab = ctx.encode(h, a ^ b)
    ctx.synchronize()
    assert torch.equal(cab, ca ^ cb)  # encoder linearity
    era = torch.empty((F, n), dtype=torch.uint8, device=dev)
    ctx.synth_erasures_uniform(83, 0, F, n, 0.19, era)  # MP-only and ML frames mixed
    da, _, _, sta = ctx.decode(h, ca, era)
    db, _, _, _ = ctx.decode(h, cb, era)
    dab, _, _, _ = ctx.decode(h, ca ^ cb, era)
    ctx.synchronize()
    ok = sta <= 1
    assert bool(ok.any()) and torch.equal(dab[ok], (da ^ db)[ok])
    assert torch.equal(da[ok], ca[ok])
    again, sw, res, st = ctx.decode(h, da, torch.zeros_like(era))
    ctx.synchronize()
    assert torch.equal(again, da) and int(sw.max()) == 1 and int(st.max()) == 0
    ctx.close()


def test_inplace_decode_extension(code_a):
    """LDPC_AMD_INPLACE (out == sym): only erased symbols are written; result equals the out-of-place decode,
    including frames that go through the ML stage."""
    torch = pytest.importorskip("torch")
    ctx = api.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    F, S, n, k = 64, 1024, code_a.n, code_a.k
    dev = torch.device("cuda", 0)
    src = torch.empty((F, k, S), dtype=torch.uint8, device=dev)
    ctx.synth_source(91, 0, F, k, S, src)
    cw = ctx.encode(h, src)
    era = torch.empty((F, n), dtype=torch.uint8, device=dev)
    ctx.synth_erasures_uniform(92, 0, F // 2, n, 0.10, era[: F // 2])
    ctx.synth_erasures_uniform(93, 0, F // 2, n, 0.21, era[F // 2:])
    sym = cw.clone()
    sym[era.bool()] = 0xC3
    ref, sw, res, st = ctx.decode(h, sym, era)
    buf = sym.clone()
    out, sw2, res2, st2 = ctx.decode(h, buf, era, inplace=True)
    ctx.synchronize()
    assert out.data_ptr() == buf.data_ptr()
    assert torch.equal(buf, ref) and torch.equal(sw, sw2) and torch.equal(st, st2)
    assert int((st == 1).sum()) > 0  # the ML stage was exercised
    with pytest.raises(api.LdpcAmdError):
        ctx.decode(h, np.zeros((1, n, 16), np.uint8), np.zeros((1, n), np.uint8), inplace=True)  # host pointers
    ctx.close()


def test_cpp_host_harness_passes():
    """ldpc_erasure_codes_amd/host/main.cpp: init_opencl -> run -> verify_output -> cleanup with the reference's CLI."""
    exe = os.path.join(ROOT, "ldpc_erasure_codes_amd", "host", "ldpc_erasure_decoder_host")
    if not os.path.exists(exe):
        pytest.skip("host harness not built")
    r = subprocess.run([exe, "-h", "-c", "1", "-p", "9", "-n", "2000", "-i", "50"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "PASSED" in r.stdout and "FAILED" not in r.stdout
    assert "frame error rate" in r.stdout and "frames/sec" in r.stdout
    # the reference harness's throughput line (OpenCL/host/src/main.cpp:652-655) is kept for scripts that parse it -- annotated,
    # because this run moves erasure patterns only
    tl = [ln for ln in r.stdout.splitlines() if ln.startswith("The throughput in information bits/sec: ")]
    assert len(tl) == 1 and "pattern-only equivalent" in tl[0] and float(tl[0].split("bits/sec: ")[1].split()[0]) > 0
    # the paper's N_T = 1e6 at PER 12/64 (Table I, tex:207: BLER 0.02, RS 7.3e-3) through the CLI: the run is streamed
    r = subprocess.run([exe, "-h", "-c", "1", "-p", "12", "-n", "1000000", "-i", "50"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = [ln for ln in r.stdout.splitlines() if "frame error rate" in ln][0]
    fer = float(line.split("frame error rate is:")[1].split(",")[0])
    rs = float(line.split("RS FER=")[1])
    assert 0.015 < fer < 0.025 and 0.0070 < rs < 0.0077, line
    r = subprocess.run([exe, "-e", "-c", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PASSED" in r.stdout


def test_wire_front_end_feeds_the_decoder(ctx, code_a):
    """SURVEY.md 8(f) rank 4 end to end: encode on the GPU -> FEC packets (header + payload) -> lossy, re-ordering
    channel -> two-buffer reassembler (batch form) -> ldpc_amd_decode_batch -> the transmitted source symbols."""
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    n, k, S, F = code_a.n, code_a.k, 16, 10
    src = synth.source(501, 0, F, k, S)
    cw = ctx.encode(h, src)
    packets = api.fec_packetize(cw, 1, 252)
    rng = np.random.default_rng(11)
    keep = [i for i in range(len(packets)) if rng.random() >= 0.08]
    order = sorted(keep, key=lambda i: i + rng.integers(0, 120))
    stream = packets[order]
    rx = api.FecRx(n, k, S)
    blocks, sym, er, used = rx.push_many(stream, F)
    tail = []
    while True:
        t = rx.flush()
        if t is None:
            break
        tail.append(t)
    assert used == len(stream) and len(blocks) + len(tail) == F
    sym = np.concatenate([sym] + [t[1][None] for t in tail])
    er = np.concatenate([er] + [t[2][None] for t in tail])
    blocks = list(blocks) + [t[0] for t in tail]
    assert blocks == [(252 + f) % 256 for f in range(F)]
    lost = np.ones((F, n), dtype=np.uint8)
    for i in keep:
        lost[i // n, i % n] = 0
    # every packet the channel lost is flagged; the only other flags are packets that arrived after their block had been
    # closed by the 'next block has > 100 packets' rule (:139) and were therefore dropped by the receiver
    assert not (lost & (er == 0)).any() and int(er.sum()) - int(lost.sum()) == rx.dropped
    out, sw, res, st = ctx.decode(h, sym, er)
    assert int(res.max()) == 0 and np.array_equal(out, cw) and np.array_equal(out[:, :k], src)
    rx.close()


def test_host_buffer_pipeline_matches_single_shot(ctx, code_a):
    """Host pointers, batch above the pipeline threshold (chunked upload / decode / download from two host threads):
    same bytes and status words as the single-shot path (LDPC_AMD_HOST_PIPELINE=0), ragged last chunk, ML frames in
    the middle of the batch."""
    h = ctx.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
    n, k, S, F = code_a.n, code_a.k, 1024, 150          # 300 MB each way: four chunks of 47/47/47/9 frames
    src = synth.source(601, 0, F, k, S)
    cw = ctx.encode(h, src)
    era = synth.erasures_uniform(602, 0, F, n, 0.10)
    era[60:70] = synth.erasures_uniform(603, 0, 10, n, 0.21)
    sym = cw.copy()
    sym[era.astype(bool)] = 0x77
    ctx.configure("LDPC_AMD_HOST_PIPELINE", 0)
    try:
        ref = ctx.decode(h, sym, era)
    finally:
        ctx.configure("LDPC_AMD_HOST_PIPELINE", None)
    got = ctx.decode(h, sym, era)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
    assert int((got[3] == 1).sum()) > 0 and np.array_equal(got[0][got[3] <= 1], cw[got[3] <= 1])


def test_two_contexts_from_two_threads(oracle, code_a):
    """SURVEY.md 8(b) threading row: a handle is self-contained (own stream, own workspaces, no library-global state
    besides the constant tables), so two handles may be driven from two host threads at the same time."""
    import threading
    results = {}

    def worker(tag, seed, per):
        c = api.Context(0)
        try:
            h = c.load_builtin_code(1, codes.DEFAULT_COEF_SEED[1])
            F = 96
            src = synth.source(seed, 0, F, code_a.k, 16)
            cw = c.encode(h, src)
            era = synth.erasures_uniform(seed + 1, 0, F, code_a.n, per)
            sym = cw.copy()
            sym[era.astype(bool)] = 0x11
            for _ in range(4):
                out, sw, res, st = c.decode(h, sym, era)
            results[tag] = (cw, out, sw, res, st, era, sym)
        finally:
            c.close()

    ts = [threading.Thread(target=worker, args=("a", 700, 0.10)), threading.Thread(target=worker, args=("b", 800, 0.20))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert set(results) == {"a", "b"}
    oc = oracle.OracleCode(code_a)
    for tag, (cw, out, sw, res, st, era, sym) in results.items():
        ok = st <= 1
        assert ok.any() and np.array_equal(out[ok], cw[ok])
        for f in (0, 17, 95):
            o_out, _, o_it, info, rc = oc.decode_packets(sym[f], era[f])
            assert sw[f] == o_it and res[f] == info[0] and np.array_equal(out[f], o_out)


def test_first_library_use_is_two_concurrent_inits(tmp_path):
    """A fresh process whose very first use of the library is two concurrent ldpc_amd_init calls (ctypes drops the GIL),
    followed by decodes with different codes and packet sizes on the two handles: the GF tables are built once (magic
    static), and the kernels' dynamic-LDS allowance is a once-only process-wide setting, not a per-launch one."""
    script = tmp_path / "first_use.py"
    script.write_text(
        "import sys, threading\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import numpy as np\n"
        "from ldpc_erasure_codes_amd import api, codes, synth\n"
        "api.load_library()\n"
        "go = threading.Barrier(2)\n"
        "bad = []\n"
        "def worker(code_ind, S, seed):\n"
        "    go.wait()\n"
        "    c = api.Context(0)\n"
        "    try:\n"
        "        c.selftest()\n"
        "        code = codes.load_builtin(code_ind)\n"
        "        h = c.load_builtin_code(code_ind, codes.DEFAULT_COEF_SEED[code_ind])\n"
        "        src = synth.source(seed, 0, 48, code.k, S)\n"
        "        src = src[:, :, 0] if S == 1 else src\n"
        "        cw = c.encode(h, src)\n"
        "        era = synth.erasures_uniform(seed + 1, 0, 48, code.n, 0.10)\n"
        "        sym = cw.copy(); sym[era.astype(bool)] = 0\n"
        "        for _ in range(6):\n"
        "            out, sw, res, st = c.decode(h, sym, era)\n"
        "            if not (np.array_equal(out, cw) and (st == 0).all()): bad.append((code_ind, S))\n"
        "    except Exception as e:\n"
        "        bad.append(repr(e))\n"
        "    finally:\n"
        "        c.close()\n"
        "ts = [threading.Thread(target=worker, args=(1, 1024, 10)), threading.Thread(target=worker, args=(2, 64, 20))]\n"
        "[t.start() for t in ts]; [t.join() for t in ts]\n"
        "print('BAD' if bad else 'GOOD', bad)\n")
    import sys
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "GOOD" in r.stdout, r.stdout + r.stderr


def test_rs_decode_rejects_malformed_positions(ctx):
    """recv_idx must be < n and strictly ascending (Matlab/ReedSolomonErasureCodes.m:80-81 builds it that way).  Host
    pointers: LDPC_AMD_EINVAL.  Device pointers: the kernels check per block and decode the offending block to zeros
    instead of indexing their tables with it; well-formed blocks of the same batch are unaffected."""
    import torch
    for (n, k, S) in ((255, 223, 1), (255, 192, 1), (255, 223, 16)):
        rs = ctx.rs_create(n, k)
        nb = 6
        src = synth.source(900 + k, 0, nb, k, S)
        src1 = src[:, :, 0] if S == 1 else src
        cw = ctx.rs_encode(rs, n, k, src1)
        rng = np.random.default_rng(5)
        idx = np.stack([np.sort(rng.choice(n, k, replace=False)) for _ in range(nb)]).astype(np.uint16)
        val = np.stack([cw[b][idx[b]] for b in range(nb)])
        good = ctx.rs_decode(rs, idx, val)
        assert np.array_equal(good, src1)
        bad_idx = idx.copy()
        bad_idx[1, 5] = n + 7                      # out of range
        bad_idx[3, 10] = bad_idx[3, 9]             # duplicate (not strictly ascending)
        bad_idx[4] = bad_idx[4][::-1]              # descending
        with pytest.raises(api.LdpcAmdError):
            ctx.rs_decode(rs, bad_idx, val)
        L = api.load_library()
        d_idx = torch.from_numpy(bad_idx.view(np.int16)).cuda()
        d_val = torch.from_numpy(val).cuda()
        d_msg = torch.full(val.shape, 0x77, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        rc = L.ldpc_amd_rs_decode_batch(ctx._h, rs, S, nb, d_idx.data_ptr(), d_val.data_ptr(), d_msg.data_ptr(), api.DEVICE_PTRS)
        assert rc == 0
        ctx.synchronize()
        msg = d_msg.cpu().numpy()
        for b in range(nb):
            if b in (1, 3, 4):
                assert not msg[b].any()
            else:
                assert np.array_equal(msg[b], src1[b])
