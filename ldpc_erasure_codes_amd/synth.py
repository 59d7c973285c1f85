"""numpy re-implementation of include/ldpc_erasure_amd_synth.h (bit-identical).

Host-side input generation for tests and the bench: GF(256) coefficients, source bytes and erasure
patterns are functions of (seed, stream, index), so the CPU oracle, this module and the HIP
generator kernels produce the same frames without shipping data.
"""
import numpy as np

STREAM_COEF = 1
STREAM_SOURCE = 2
STREAM_ERASE = 3
STREAM_BURST_E = 4
STREAM_BURST_S = 5
STREAM_RS = 6

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def u64(seed, stream, idx):
    """64 uniform bits for every element of idx (array of uint64 indices)."""
    with np.errstate(over="ignore"):
        key = _mix64(np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1342543DE82EF95)))
        return _mix64(key + np.asarray(idx, dtype=np.uint64))


def u32(seed, stream, idx):
    return (u64(seed, stream, idx) >> np.uint64(32)).astype(np.uint32)


def byte(seed, stream, idx):
    return (u64(seed, stream, idx) >> np.uint64(56)).astype(np.uint8)


def nonzero(seed, stream, idx):
    return (1 + (u64(seed, stream, idx) >> np.uint64(32)) % np.uint64(255)).astype(np.uint8)


def threshold(p):
    if p <= 0.0:
        return 0
    if p >= 1.0:
        return 1 << 32
    return int(p * 4294967296.0 + 0.5)


def coefs(seed, nnz):
    """GF(256) coefficient (1..255) of every non-zero of H, in CSR order."""
    return nonzero(seed, STREAM_COEF, np.arange(nnz, dtype=np.uint64))


def source(seed, frame0, nframes, k, S):
    base = np.uint64(frame0) * np.uint64(k) * np.uint64(S)
    idx = base + np.arange(nframes * k * S, dtype=np.uint64)
    return byte(seed, STREAM_SOURCE, idx).reshape(nframes, k, S)


def erasures_uniform(seed, frame0, nframes, n, per):
    base = np.uint64(frame0) * np.uint64(n)
    idx = base + np.arange(nframes * n, dtype=np.uint64)
    return (u32(seed, STREAM_ERASE, idx).astype(np.uint64) < np.uint64(threshold(per))).astype(np.uint8).reshape(nframes, n)


def erasures_bursty(seed, frame0, nframes, n, alpha, beta, good_transition_bias, transition=0.1):
    """Gilbert-Elliott stream (reference: Matlab/Bursty_Error_Channel_Model_Generator.m:12-47), chain
    state carried across symbols and frames from global symbol 0."""
    last = (frame0 + nframes) * n
    idx = np.arange(last, dtype=np.uint64)
    r1 = u32(seed, STREAM_BURST_E, idx).astype(np.uint64)
    r2 = u32(seed, STREAM_BURST_S, idx).astype(np.uint64)
    ta, tb = np.uint64(threshold(alpha)), np.uint64(threshold(beta))
    t10 = np.uint64(threshold(transition / good_transition_bias))
    t01 = np.uint64(threshold(transition))
    to_bad = r2 < t10   # taken when in state 0
    to_good = r2 < t01  # taken when in state 1
    state = np.zeros(last, dtype=np.uint8)
    s = 0
    tb_l, tg_l = to_bad.tolist(), to_good.tolist()
    st = [0] * last
    for i in range(last):
        st[i] = s
        if s == 0:
            if tb_l[i]:
                s = 1
        elif tg_l[i]:
            s = 0
    state[:] = st
    era = np.where(state == 0, r1 < ta, r1 < tb).astype(np.uint8)
    return era[frame0 * n:].reshape(nframes, n)


def threefry4x32_20(ctr, key):
    """Vectorised Threefry4x32-20 (include/ldpc_erasure_amd_synth.h): ctr [N,4] uint32, key [4] -> [N,4] uint32."""
    R0 = (10, 11, 13, 23, 6, 17, 25, 18)
    R1 = (26, 21, 27, 5, 20, 11, 10, 20)
    ctr = np.asarray(ctr, dtype=np.uint32).reshape(-1, 4)
    ks = [np.uint32(k) for k in key]
    ks.append(np.uint32(0x1BD11BDA) ^ ks[0] ^ ks[1] ^ ks[2] ^ ks[3])

    def rotl(x, r):
        return (x << np.uint32(r)) | (x >> np.uint32(32 - r))

    with np.errstate(over="ignore"):
        x = [ctr[:, i] + ks[i] for i in range(4)]
        for r in range(20):
            if r % 2 == 0:
                x[0] = x[0] + x[1]; x[1] = rotl(x[1], R0[r % 8]) ^ x[0]
                x[2] = x[2] + x[3]; x[3] = rotl(x[3], R1[r % 8]) ^ x[2]
            else:
                x[0] = x[0] + x[3]; x[3] = rotl(x[3], R0[r % 8]) ^ x[0]
                x[2] = x[2] + x[1]; x[1] = rotl(x[1], R1[r % 8]) ^ x[2]
            if r % 4 == 3:
                s_ = r // 4 + 1
                for i in range(4):
                    x[i] = x[i] + ks[(s_ + i) % 5]
                x[3] = x[3] + np.uint32(s_)
    return np.stack(x, axis=1)


def fpga_erasures(seed, per64, nframes, n):
    """Erasure flags of the FPGA data_in kernel (OpenCL/device/ldpc_erasure_decoder_top.cl:74-110)."""
    g = np.arange(nframes * n, dtype=np.uint64)
    ctr = np.zeros((g.size, 4), dtype=np.uint32)
    ctr[:, 0] = ((g + np.uint64(1)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    out = threefry4x32_20(ctr, (1, np.uint32(seed & 0xFFFFFFFF), 0, 0))
    return ((out[:, 0] & np.uint32(0x3F)) < np.uint32(max(per64, 0))).astype(np.uint8).reshape(nframes, n)
