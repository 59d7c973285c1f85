#!/usr/bin/env python3
"""Freeze decoder input/output vectors produced by the CPU oracle into tests/golden/ldpc_golden_vectors.npz.

The reference ships no decoder I/O vectors, so these are NOT reference-produced: they are
the oracle's outputs on seeded inputs, committed so that (a) the oracle cannot drift silently and (b) the GPU
path can be checked on the GPU box without regenerating anything.  Cases (SURVEY.md 8c list): no erasure, one
erasure, 2 / 3 / >=5 sweeps, sweep cap + ML success, rank-deficient ML, more erasures than checks, all parity
erased, bursty pattern; plus two S = 16 packet frames and RS(255,223) blocks.

    python tools/make_golden_vectors.py        # rewrites the fixture
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ldpc_erasure_codes_amd import codes, synth  # noqa: E402
from oracle import oracle_py  # noqa: E402


def main():
    code = codes.load_builtin(1)  # (2040,1530), H_nb seed 2040
    oc = oracle_py.OracleCode(code)
    n, k, m = code.n, code.k, code.m
    pats = []
    names = []

    def add(name, era):
        names.append(name)
        pats.append(era.astype(np.uint8))

    z = np.zeros(n, np.uint8)
    add("no_erasure", z)
    e = z.copy(); e[17] = 1; add("one_erasure", e)
    e = z.copy(); e[k:] = 1; add("all_parity_erased", e)
    e = z.copy(); e[:600] = 1; add("more_erasures_than_checks", e)
    e = z.copy(); e[100:100 + m] = 1; add("exactly_n_minus_k", e)
    for i, per in enumerate((0.10, 0.10, 0.1406, 0.18, 0.19, 0.20, 0.205, 0.21, 0.215, 0.22, 0.225, 0.23, 0.235, 0.24, 0.245)):
        add(f"uniform_{per}", synth.erasures_uniform(9000 + i, 0, 1, n, per)[0])
    b = synth.erasures_bursty(77, 0, 4, n, 0.12, 0.8, 10.0)
    for i in range(4):
        add(f"bursty_{i}", b[i])
    era = np.stack(pats)
    F = era.shape[0]
    src = synth.source(424242, 0, F, k, 1)[:, :, 0]
    cw = np.stack([oc.encode(s) for s in src])
    sym = cw.copy()
    sym[era.astype(bool)] = 0xA5
    out, sw, res, st = oc.decode_batch_s1(sym, era)
    print("status histogram:", np.bincount(st, minlength=4).tolist(), "sweeps:", sw.tolist())
    assert set(np.unique(st)) == {0, 1, 2, 3}, "fixture must cover every status"

    # packets: S = 16, one MP-only frame and one ML frame
    S = 16
    psrc = synth.source(515151, 0, 2, k, S)
    pcw = np.stack([oc.encode(psrc[f]) for f in range(2)])
    pera = np.stack([synth.erasures_uniform(9100, 0, 1, n, 0.12)[0], synth.erasures_uniform(9101, 0, 1, n, 0.215)[0]])
    psym = pcw.copy()
    psym[pera.astype(bool)] = 0x3C
    pout, psw, pres = [], [], []
    for f in range(2):
        o, _, it, info, rc = oc.decode_packets(psym[f], pera[f])
        pout.append(o); psw.append(it); pres.append(int(info[0]))

    # RS(255,223): 8 blocks, assorted erasure counts
    rn, rk = 255, 223
    G = oracle_py.rs_generator(rn, rk)
    rng = np.random.default_rng(2026)
    rsrc = rng.integers(0, 256, size=(8, rk)).astype(np.uint8)
    ridx = np.zeros((8, rk), np.uint16)
    rval = np.zeros((8, rk), np.uint8)
    for b_ in range(8):
        cwb = oracle_py.rs_encode(G, rsrc[b_])
        ne = [0, 1, 32, 31, 16, 7, 32, 20][b_]
        ridx[b_] = np.sort(rng.permutation(rn)[: rn - ne])[:rk]
        rval[b_] = cwb[ridx[b_]]
        msg, rc = oracle_py.rs_decode(G, ridx[b_], rval[b_])
        assert rc == 0 and np.array_equal(msg, rsrc[b_])

    path = os.path.join(ROOT, "tests", "golden", "ldpc_golden_vectors.npz")
    np.savez_compressed(
        path, names=np.array(names), code_ind=1, coef_seed=codes.DEFAULT_COEF_SEED[1],
        erased_bits=np.packbits(era, axis=1), sym=sym, out=out, sweeps=sw, residual=res, status=st, codeword=cw,
        p_S=S, p_erased_bits=np.packbits(pera, axis=1), p_sym=psym, p_out=np.stack(pout), p_sweeps=np.array(psw),
        p_residual=np.array(pres), rs_n=rn, rs_k=rk, rs_idx=ridx, rs_val=rval, rs_msg=rsrc)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
