#!/usr/bin/env python3
"""CPU model of the ML stage's FAST PATH ("peel and inactivate"), the design note of csrc/ml_pi.inc in executable form.

The reference solves the residual system H(:, erased) x = rhs by Gaussian elimination in its own pivot order
(Matlab/My_LDPC_HybridML_NonBinary_Erasure_Decoder.m:61-128).  When that system has full column rank its solution is UNIQUE, so
any exact solver returns the reference's bytes; only rank-deficient frames (the elimination breaks off, :87-90, and the partial
right-hand sides are still written back, :127) need the reference's order.  The fast path therefore
  1. keeps peeling the residual graph (most residual systems of the 10-sweep cap are not stopping sets at all), and when no
     check has a single unknown left it INACTIVATES one unknown -- treats it as a symbol z_j to be found later -- and goes on;
  2. carries, for every check it used, the coefficients of its value in z (one byte per inactive unknown);
  3. solves the small dense system the unused checks give for z (Gauss-Jordan, I x I, I = number of inactivations);
  4. corrects the peeled values with z.
and hands a frame back to the exact elimination when the dense system is singular (rank-deficient frame) or I > 256.

This file builds the same SOLVE SCHEDULE the device code emits -- levels of ops `slot[dst] ^= coef * source` -- runs it on
bytes and compares with oracle/ (tests/test_pi_model_cpu.py).  It is test infrastructure, not product code."""
import numpy as np

EXP = np.zeros(1024, dtype=np.int64)
LOG = np.zeros(256, dtype=np.int64)


def _init_tables():
    x = 1
    for i in range(255):
        EXP[i] = x
        LOG[x] = i
        x <<= 1
        if x & 0x100:
            x ^= 0x171
    for i in range(255, 1024):
        EXP[i] = EXP[i % 255]


_init_tables()
MUL = np.zeros((256, 256), dtype=np.uint8)
for _a in range(1, 256):
    MUL[_a, 1:] = EXP[(LOG[_a] + LOG[1:256]) % 255]
INV = np.zeros(256, dtype=np.uint8)
INV[1:] = EXP[(255 - LOG[1:256]) % 255]

IMAX = 256


class NeedExactPath(Exception):
    pass


def build_schedule(code, erased, verify=True):
    """erased: bool[n], the symbols still unknown after message passing.  Returns (levels, nslots, info): levels[L] = list of ops
    (dst, coef, src); level 0: dst slot, src symbol; last level: dst symbol, src slot; others slot <- slot."""
    n, m = code.n, code.n - code.k
    rp, cols, coefs = code.row_ptr, code.cols, code.coefs
    rows = [(cols[rp[r]:rp[r + 1]].astype(int), coefs[rp[r]:rp[r + 1]].astype(int)) for r in range(m)]
    colrows = {}
    for r in range(m):
        for c, h in zip(*rows[r]):
            if erased[c]:
                colrows.setdefault(int(c), []).append(r)
    UNK, KNOWN = -2, -1
    vinfo = np.where(erased, UNK, KNOWN).astype(int)       # >= 0: peel step; <= -10: inactive index -(10 + j)
    cnt = np.array([int(erased[rows[r][0]].sum()) for r in range(m)])
    touched = cnt > 0
    used = np.zeros(m, dtype=bool)
    E = int(erased.sum())
    steps = []          # (check, var)
    slvl = []           # level of the step (0: no peeled neighbour)
    inact = []
    remaining = E
    while remaining:
        ones = [r for r in range(m) if cnt[r] == 1 and not used[r]]
        if ones:
            claimed = {}
            for r in ones:                      # one round: every check with a single unknown claims it, first claim wins
                c_, _ = rows[r]
                v = [int(c) for c in c_ if vinfo[c] == UNK]
                if not v:
                    continue
                v = v[0]
                if v in claimed:
                    continue
                claimed[v] = r
            for v, r in claimed.items():
                lev = 0
                for c in rows[r][0]:
                    if c != v and vinfo[c] >= 0:
                        lev = max(lev, slvl[vinfo[c]] + 1)
                vinfo[v] = len(steps)
                steps.append((r, v)); slvl.append(lev)
                used[r] = True
                for rr in colrows[v]:
                    cnt[rr] -= 1
                remaining -= 1
        else:
            cand = [r for r in range(m) if cnt[r] >= 2 and not used[r]]
            r = min(cand, key=lambda q: (cnt[q], q))
            v = [int(c) for c in rows[r][0] if vinfo[c] == UNK][0]
            if len(inact) >= IMAX:
                raise NeedExactPath("more than %d inactivations" % IMAX)
            vinfo[v] = -(10 + len(inact))
            inact.append(v)
            for rr in colrows[v]:
                cnt[rr] -= 1
            remaining -= 1
    P, I = len(steps), len(inact)
    ginv = [int(INV[[h for c, h in zip(*rows[r]) if c == v][0]]) for r, v in steps]

    def symbolic(r, skip):
        a = np.zeros(I, dtype=np.uint8)
        for c, h in zip(*rows[r]):
            if c == skip:
                continue
            vi = vinfo[c]
            if vi >= 0:
                a ^= MUL[MUL[h, ginv[vi]]][AV[steps[vi][0]]]
            elif vi <= -10:
                a[-vi - 10] ^= h
        return a
    AV = {}
    if I:
        for r, v in steps:
            AV[r] = symbolic(r, v)
    # dense part: Gauss-Jordan on the unused touched checks, multipliers kept in the bytes they zero
    sel = []            # pivot row of column j
    diag = []
    if I:
        cand = [r for r in range(m) if touched[r] and not used[r]]
        T = {r: symbolic(r, -1) for r in cand}
        pivoted = set()
        for j in range(I):
            pr = next((r for r in cand if r not in pivoted and T[r][j]), None)
            if pr is None:
                raise NeedExactPath("rank-deficient")
            pivoted.add(pr)
            sel.append(pr)
            d = int(T[pr][j]); diag.append(d)
            for r in cand:
                if r == pr or not T[r][j]:
                    continue
                c = int(MUL[T[r][j], INV[d]])
                T[r][j + 1:] ^= MUL[c][T[pr][j + 1:]]
                T[r][j] = c                                    # the multiplier takes the byte it zeroed
    # ---- the schedule
    slot_of_check = {r: t for t, (r, v) in enumerate(steps)}
    for j, r in enumerate(sel):
        slot_of_check[r] = P + j
    dA = max(slvl) if slvl else 0
    sel_lvl = []
    for r in sel:
        lev = 0
        for c in rows[r][0]:
            if vinfo[c] >= 0:
                lev = max(lev, slvl[vinfo[c]] + 1)
        sel_lvl.append(lev)
    base = max([dA] + sel_lvl) + 1 if I else dA + 1          # first Gauss-Jordan level
    red = [r for r in range(m) if touched[r] and not used[r] and r not in sel] if verify else []
    Lv = base + (I + 1 if I else 0)                          # verified: the level that completes the redundant checks
    NL = Lv + (1 if red else 0) + 1
    levels = [[] for _ in range(NL)]
    userows = [(r, v, slvl[t]) for t, (r, v) in enumerate(steps)] + [(r, -1, sel_lvl[j]) for j, r in enumerate(sel)]
    for r, v, lev in userows:
        d = slot_of_check[r]
        for c, h in zip(*rows[r]):
            if c == v:
                continue
            vi = vinfo[c]
            if vi == KNOWN:
                levels[0].append((d, int(h), int(c)))
            elif vi >= 0:
                levels[lev].append((d, int(MUL[h, ginv[vi]]), vi))
    for j in range(I):
        for i, r in enumerate(sel):
            if i != j and T[r][j]:
                levels[base + j].append((P + i, int(T[r][j]), P + j))
    if I:
        for t, (r, v) in enumerate(steps):
            for j in range(I):
                if AV[r][j]:
                    levels[base + I].append((t, int(MUL[AV[r][j], INV[diag[j]]]), P + j))
    # verified: the touched checks the solution did not use must hold too -- else the system is inconsistent (the received symbols
    # were not a codeword) and the reference's bytes depend on ITS choice of equations: the frame goes to the exact elimination
    for k_, r in enumerate(red):
        q = P + I + k_
        for c, h in zip(*rows[r]):
            vi = vinfo[c]
            if vi == KNOWN:
                levels[0].append((q, int(h), int(c)))
            elif vi >= 0:
                levels[Lv].append((q, int(MUL[h, ginv[vi]]), vi))
            else:
                j = -vi - 10
                levels[Lv].append((q, int(MUL[h, INV[diag[j]]]), P + j))
        levels[NL - 1].append((0xFFFF, 1, q))
    for t, (r, v) in enumerate(steps):
        levels[NL - 1].append((v, ginv[t], t))
    for j, v in enumerate(inact):
        levels[NL - 1].append((v, int(INV[diag[j]]), P + j))
    return levels, P + I + len(red), dict(E=E, P=P, I=I, depth=dA, NL=NL, ops=sum(len(x) for x in levels), ops0=len(levels[0]))


def run_schedule(levels, nslots, out):
    """out uint8[n, S]: known symbols valid; the unknown rows are overwritten.  Returns (out, consistent): consistent = every check op
    of the last level found a zero slot (always True for a schedule built with verify=False)."""
    S = out.shape[1]
    slots = np.zeros((nslots, S), dtype=np.uint8)
    for d, c, s in levels[0]:
        slots[d] ^= MUL[c][out[s]]
    for L in range(1, len(levels) - 1):
        dsts = {d for d, _, _ in levels[L]}
        for d, c, s in levels[L]:
            assert s not in dsts, "a level reads a slot it writes"
            slots[d] ^= MUL[c][slots[s]]
    consistent = True
    for d, c, s in levels[-1]:
        if d == 0xFFFF:
            consistent = consistent and not slots[s].any()
        else:
            out[d] = MUL[c][slots[s]]
    return out, consistent
