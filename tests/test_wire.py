"""Host-side wire format (include/ldpc_erasure_amd_wire.h): FEC header, packetiser and the
two-buffer reassembler.  CPU only.  The reference for the receiver is a draft that does not compile
(OpenCL/device/ldpc_erasure_decoder_with_reordering_logic.cl), so the checker here is a second restatement of the same
lines written independently in Python -- parity is pinned to the header arithmetic of the encoder kernel and to the
draft's evident control flow, not to reference-produced vectors (none exist)."""
import numpy as np
import pytest

from ldpc_erasure_codes_amd import api


# ---------------------------------------------------------------- second restatement (test infrastructure)
def header_word(fec_class, block, symbol):
    # OpenCL/device/ldpc_erasure_encoder_VITA_in_UDP_out.cl:112-113 / :175-176
    dout = 0x00000000ffffffff & (((fec_class & 0xff) << 24) | ((block & 0xff) << 16) | (symbol & 0xffff))
    return ((dout << 32) & 0xffffffff00000000) | (dout & 0x00000000ffffffff)


class DraftReceiver:
    """OpenCL/device/ldpc_erasure_decoder_with_reordering_logic.cl:44-141,214-243, one packet per loop pass."""

    def __init__(self, n, k, S):
        self.n, self.k, self.S = n, k, S
        self.codeword = [self._empty(), self._empty()]          # codeword_first, codeword_second
        self.is_first_codeword = 1
        self.cur_block_num = self.next_block_num = -1
        self.cur_block_num_cnt = self.next_block_num_cnt = 0
        self.desired_parity_rx = int(np.floor((n - k) * 0.8 + 0.5))   # round(), :54
        self.min_parity_rx = int(np.floor((n - k) * 0.2 + 0.5))       # :55
        self.dropped = 0

    def _empty(self):
        return [np.zeros((self.n, self.S), dtype=np.uint8), np.ones(self.n, dtype=np.uint8)]

    def push(self, packet):
        hdr = int.from_bytes(bytes(packet[:8]), "little")
        blockNum, symbolNum = (hdr >> 16) & 0xff, hdr & 0xffff        # :83-85
        if self.cur_block_num == -1 and self.next_block_num == -1:   # :88-91
            self.cur_block_num = blockNum
            self.next_block_num = (blockNum + 1) % 256
        cur = 0 if self.is_first_codeword == 1 else 1
        if symbolNum >= self.n:
            self.dropped += 1
        elif blockNum == self.cur_block_num:                          # :98-105,120-127
            self.codeword[cur][0][symbolNum] = packet[8:]
            self.codeword[cur][1][symbolNum] = 0
            self.cur_block_num_cnt += 1
        elif blockNum == self.next_block_num:                         # :107-114,129-136
            self.codeword[1 - cur][0][symbolNum] = packet[8:]
            self.codeword[1 - cur][1][symbolNum] = 0
            self.next_block_num_cnt += 1
        else:
            self.dropped += 1
        if (self.cur_block_num_cnt == self.n
                or (self.cur_block_num_cnt > self.k + self.desired_parity_rx and self.next_block_num_cnt > 10)
                or (self.cur_block_num_cnt > self.k + self.min_parity_rx and self.next_block_num_cnt > 100)):   # :139
            return self._close()
        return None

    def _close(self):
        cur = 0 if self.is_first_codeword == 1 else 1
        out = (self.cur_block_num, self.codeword[cur][0].copy(), self.codeword[cur][1].copy())
        self.cur_block_num = self.next_block_num                      # :216-219
        self.next_block_num = (self.next_block_num + 1) % 256
        self.cur_block_num_cnt = self.next_block_num_cnt
        self.next_block_num_cnt = 0
        self.codeword[cur] = self._empty()                            # :220-240
        self.is_first_codeword = 0 if self.is_first_codeword else 1   # :241
        return out

    def flush(self):
        if self.cur_block_num == -1 or (self.cur_block_num_cnt == 0 and self.next_block_num_cnt == 0):
            return None
        return self._close()


# ---------------------------------------------------------------- tests
def test_fec_header_known_answers():
    # {class 1 | block 0x12 | symbol 0x0345} -> 0x01120345 in both halves
    assert api.fec_header_pack(1, 0x12, 0x0345) == 0x0112034501120345 == header_word(1, 0x12, 0x0345)
    assert api.fec_header_pack(0x1FF, 0x1FF, 0x1FFFF) == 0xFFFFFFFFFFFFFFFF      # fields are masked, not range-checked
    assert api.fec_header_unpack(0x0112034501120345) == (1, 0x12, 0x0345)
    rng = np.random.default_rng(1)
    for _ in range(200):
        c, b, s = (int(x) for x in rng.integers(0, 1 << 20, size=3))
        w = api.fec_header_pack(c, b, s)
        assert w == header_word(c, b, s) and api.fec_header_unpack(w) == (c & 0xff, b & 0xff, s & 0xffff)


def test_packetize_layout():
    rng = np.random.default_rng(2)
    frames = rng.integers(0, 256, size=(3, 7, 16), dtype=np.uint8)
    p = api.fec_packetize(frames, fec_class=1, block0=254)
    assert p.shape == (21, 24)
    for f in range(3):
        for j in range(7):
            row = p[f * 7 + j]
            assert int.from_bytes(bytes(row[:8]), "little") == header_word(1, 254 + f, j)   # block wraps 254, 255, 0
            assert np.array_equal(row[8:], frames[f, j])


def _channel(packets, n, rng, loss, window):
    """Drops packets with probability `loss` and delivers the rest out of order inside a sliding window."""
    keep = [i for i in range(len(packets)) if rng.random() >= loss]
    order = sorted(keep, key=lambda i: i + rng.integers(0, window))
    return [packets[i] for i in order]


@pytest.mark.parametrize("n,k,S,loss,window,dup", [(300, 200, 16, 0.1, 1, 0.0), (300, 200, 16, 0.15, 25, 0.02),
                                                    (400, 300, 32, 0.08, 90, 0.0), (2040, 1530, 16, 0.1, 300, 0.001)])
def test_reassembler_matches_second_restatement(n, k, S, loss, window, dup):
    rng = np.random.default_rng(n + window)
    F = 270 if n < 1000 else 6       # more than 256 blocks: the 8-bit block field wraps
    frames = rng.integers(0, 256, size=(F, n, S), dtype=np.uint8)
    packets = api.fec_packetize(frames, 1, 250)
    stream = _channel(packets, n, rng, loss, window)
    stream = [p for q in stream for p in ([q, q] if rng.random() < dup else [q])]     # duplicates count twice, as in the draft
    rx, ref = api.FecRx(n, k, S), DraftReceiver(n, k, S)
    got, want = [], []
    for pkt in stream:
        a, b = rx.push(pkt), ref.push(pkt)
        assert (a is None) == (b is None)
        if a is not None:
            got.append(a); want.append(b)
    while True:
        a, b = rx.flush(), ref.flush()
        assert (a is None) == (b is None)
        if a is None:
            break
        got.append(a); want.append(b)
    # (the rules of :139 use absolute packet counts -- 10 and 100 of the NEXT block -- so they only make progress on a
    #  lossy stream when n is well above 100; the codes of the reference have n >= 2000)
    assert len(got) >= F - 2 and rx.dropped == ref.dropped
    for (ba, sa, ea), (bb, sb, eb) in zip(got, want):
        assert ba == bb and np.array_equal(sa, sb) and np.array_equal(ea, eb)
    # what comes out is usable by the decoder: received symbols carry their payload, erased ones are zero
    for i, (blk, sym, er) in enumerate(got[:5]):
        f = (blk - 250) % 256 if F <= 256 else None
        if f is not None:
            assert np.array_equal(sym[er == 0], frames[f][er == 0]) and not sym[er == 1].any()


def test_decode_start_rules():
    """Each clause of :139 on its own: a full block; > k + 0.8(n-k) symbols once 11 packets of the next block are in;
    > k + 0.2(n-k) symbols once 101 are in; stale blocks are dropped."""
    n, k, S = 300, 200, 16
    z = np.zeros(S, dtype=np.uint8)

    def pkt(block, sym):
        return np.concatenate([np.frombuffer(header_word(1, block, sym).to_bytes(8, "little"), dtype=np.uint8), z])

    rx = api.FecRx(n, k, S)
    for j in range(n - 1):
        assert rx.push(pkt(7, j)) is None
    blk, _, er = rx.push(pkt(7, n - 1))
    assert blk == 7 and not er.any()
    # second rule: 281 symbols of block 8 (> 200 + 80) and 11 of block 9
    for j in range(281):
        assert rx.push(pkt(8, j)) is None
    for j in range(10):
        assert rx.push(pkt(9, j)) is None
    assert rx.push(pkt(5, 0)) is None and rx.dropped == 1           # neither current nor next
    blk, _, er = rx.push(pkt(9, 10))
    assert blk == 8 and int(er.sum()) == n - 281
    # third rule: block 9 already holds 11 packets; bring it to 221 (> 200 + 20), then 101 packets of block 10
    for j in range(11, 221):
        assert rx.push(pkt(9, j)) is None
    for j in range(100):
        assert rx.push(pkt(10, j)) is None
    blk, _, er = rx.push(pkt(10, 100))
    assert blk == 9 and int(er.sum()) == n - 221
    assert rx.push(pkt(10, 400)) is None and rx.dropped == 2          # symbol number beyond n
    blk, _, er = rx.flush()
    assert blk == 10 and int((er == 0).sum()) == 101
    assert rx.flush() is None
