"""Second, independent restatement of the reference's Matlab functions -- pure Python, 1-BASED indexing,
Matlab's own data structures (dense H, Vlist with the degree in column 1, -1 for erasures) and the
reference's OWN lookup tables (tests/golden/gf256_tables_ref.npz == Matlab/GF_256_add_mult_inv_tables.mat).

It exists only to cross-check oracle/oracle.c (a transliteration error would have to be made twice, in
two different styles, to go unnoticed).  Slow: use on small codes / few frames.

Each function cites the reference file it follows (paths relative to /root/reference).
"""
import os

import numpy as np

_G = np.load(os.path.join(os.path.dirname(__file__), "golden", "gf256_tables_ref.npz"))
GF_add_lookup = _G["GF_add_lookup"].astype(int)
GF_mult_lookup = _G["GF_mult_lookup"].astype(int)
GF_inv_lookup = _G["GF_inv_lookup"].astype(int)


class M1:
    """1-based vector/matrix view over a numpy int array (Matlab style A(i), A(i,j))."""

    def __init__(self, a):
        self.a = np.array(a, dtype=int)

    def __getitem__(self, ix):
        if isinstance(ix, tuple):
            return int(self.a[ix[0] - 1, ix[1] - 1])
        return int(self.a[ix - 1])

    def __setitem__(self, ix, v):
        if isinstance(ix, tuple):
            self.a[ix[0] - 1, ix[1] - 1] = v
        else:
            self.a[ix - 1] = v


def add(a, b):
    return int(GF_add_lookup[a, b])  # GF_add_lookup(a+1, b+1)


def mult(a, b):
    return int(GF_mult_lookup[a, b])  # GF_mult_lookup(a+1, b+1)


def inv(x):
    return int(GF_inv_lookup[x - 1])  # GF_inv_lookup(x)


def build_vlist(H):
    """Matlab/ErasureCodes_NonBinaryLDPCSim.m:91-107 (H: dense m x n of GF(256) coefficients)."""
    Hb = (H != 0).astype(int)
    m, n = H.shape
    width = int(Hb.sum(axis=1).max()) + 1
    Vlist = M1(np.zeros((m, width)))
    Vlist_val = M1(np.zeros((m, width)))
    for jj in range(1, m + 1):
        Vlist[jj, 1] = int(Hb[jj - 1].sum())
        Vlist_val[jj, 1] = Vlist[jj, 1]
        icnt = 0
        for ii in range(1, n + 1):
            if Hb[jj - 1, ii - 1] == 1:
                icnt += 1
                Vlist[jj, icnt + 1] = ii
                Vlist_val[jj, icnt + 1] = int(H[jj - 1, ii - 1])
    return Vlist, Vlist_val


def encode(H, source_vec):
    """Matlab/ErasureCodes_NonBinaryLDPCSim.m:173-182."""
    m, n = H.shape
    k = n - m
    Vlist, Vlist_val = build_vlist(H)
    cw = M1(np.zeros(n))
    for i in range(1, k + 1):
        cw[i] = int(source_vec[i - 1])
    for pp in range(1, n - k + 1):
        gf_sum = 0
        for ll in range(1, Vlist_val[pp, 1] - 1 + 1):
            gf_sum = add(gf_sum, mult(cw[Vlist[pp, ll + 1]], Vlist_val[pp, ll + 1]))
        cw[k + pp] = mult(gf_sum, inv(Vlist_val[pp, Vlist_val[pp, 1] + 1]))
    return cw.a.copy()


def hybridml_nonbinary_decode(recv_vec_val, H, itenum=10, do_ML_decode=1):
    """Matlab/My_LDPC_HybridML_NonBinary_Erasure_Decoder.m:4-130.
    Returns (Msg, iterations, dont_do_jordan or None)."""
    m, n = H.shape
    k = n - m
    H_sparse = M1(H)
    Vlist, _ = build_vlist(H)
    y_current = M1(recv_vec_val)
    stopsig, itestep = 0, 0
    num_cur_erasures = 0
    while stopsig == 0 and itestep < itenum:
        itestep += 1
        for ii in range(1, m + 1):
            num_erasures, erasure_ind = 0, 0
            for jj in range(1, Vlist[ii, 1] + 1):
                if y_current[Vlist[ii, jj + 1]] == -1:
                    num_erasures += 1
                    erasure_ind = Vlist[ii, jj + 1]
            if num_erasures == 1:
                neigh = [Vlist[ii, j] for j in range(2, Vlist[ii, 1] + 2)]
                check_indices = sorted(set(neigh) ^ {erasure_ind})  # setxor
                gf_sum = 0
                for kk in range(1, len(check_indices) + 1):
                    ci = check_indices[kk - 1]
                    gf_sum = add(gf_sum, mult(y_current[ci], H_sparse[ii, ci]))
                y_current[erasure_ind] = mult(gf_sum, inv(H_sparse[ii, erasure_ind]))
        num_cur_erasures = int(np.sum(y_current.a == -1))
        if num_cur_erasures == 0:
            stopsig = 1
    djordan = None
    if num_cur_erasures > 0 and do_ML_decode == 1:
        erasure_ind = [j for j in range(1, n + 1) if y_current[j] == -1]
        num_erasures = len(erasure_ind)
        find_inv = M1(H[:, [e - 1 for e in erasure_ind]])
        non_erasure_ind = set(range(1, n + 1)) - set(erasure_ind)
        rhs = M1(np.zeros(n - k))
        for kk in range(1, n - k + 1):
            neigh = [Vlist[kk, j] for j in range(2, Vlist[kk, 1] + 2)]
            non_erasure_ind_kk = sorted(set(neigh) & non_erasure_ind)  # intersect
            gf_sum = 0
            for ll in non_erasure_ind_kk:
                gf_sum = add(gf_sum, mult(y_current[ll], H_sparse[kk, ll]))
            rhs[kk] = gf_sum
        dont_do_jordan = 0
        for col in range(1, num_erasures + 1):
            non_zero_ind = [r for r in range(col, m + 1) if find_inv[r, col] != 0]
            if len(non_zero_ind) == 0:
                dont_do_jordan = 1
                break
            p = non_zero_ind[0]
            rhs[col], rhs[p] = rhs[p], rhs[col]
            tmp = find_inv.a[col - 1, :].copy()
            find_inv.a[col - 1, :] = find_inv.a[p - 1, :]
            find_inv.a[p - 1, :] = tmp
            non_zero_indices = [c for c in range(1, num_erasures + 1) if find_inv[col, c] != 0]
            multiplier = inv(find_inv[col, non_zero_indices[0]])
            find_inv[col, col] = mult(find_inv[col, col], multiplier)
            rhs[col] = mult(rhs[col], multiplier)
            for kk in range(2, len(non_zero_indices) + 1):
                c = non_zero_indices[kk - 1]
                find_inv[col, c] = mult(find_inv[col, c], multiplier)
            for ii in range(2, len(non_zero_ind) + 1):
                r = non_zero_ind[ii - 1]
                nzr = sorted({c for c in range(1, num_erasures + 1) if find_inv[r, c] != 0}
                             | {c for c in range(1, num_erasures + 1) if find_inv[col, c] != 0})  # union
                multiplier = find_inv[r, col]
                for c in nzr:
                    find_inv[r, c] = add(find_inv[r, c], mult(multiplier, find_inv[col, c]))
                rhs[r] = add(rhs[r], mult(multiplier, rhs[col]))
        if not dont_do_jordan:
            for col in range(num_erasures, 1, -1):
                non_zero_ind = [r for r in range(1, col) if find_inv[r, col] != 0]
                for r in non_zero_ind:
                    rhs[r] = add(rhs[r], mult(find_inv[r, col], rhs[col]))
                    find_inv[r, col] = 0
        for t in range(1, num_erasures + 1):
            y_current[erasure_ind[t - 1]] = rhs[t]
        djordan = dont_do_jordan
    return y_current.a.copy(), itestep, djordan


def binary_mp_decode(recv_vec_val, Hb, itenum=50):
    """Matlab/My_LDPC_Erasure_Decoder.m:3-50."""
    m, n = Hb.shape
    Vlist, _ = build_vlist(Hb)
    y = M1(recv_vec_val)
    stopsig, itestep = 0, 0
    while stopsig == 0 and itestep < itenum:
        itestep += 1
        for ii in range(1, m + 1):
            num_erasures, erasure_ind = 0, 0
            for jj in range(1, Vlist[ii, 1] + 1):
                if y[Vlist[ii, jj + 1]] == -1:
                    num_erasures += 1
                    erasure_ind = Vlist[ii, jj + 1]
            if num_erasures == 1:
                neigh = [Vlist[ii, j] for j in range(2, Vlist[ii, 1] + 2)]
                others = sorted(set(neigh) ^ {erasure_ind})
                y[erasure_ind] = sum(y[c] for c in others) % 2
        if int(np.sum(y.a == -1)) == 0:
            stopsig = 1
    return y.a.copy(), itestep


def rs_generator(n, k):
    """Matlab/Test_My_RS_Decode.m:22,30-37 (alpha = 2; powers through repeated table multiplication)."""
    def gpow(a, e):
        r = 1
        for _ in range(e):
            r = mult(r, a)
        return r

    alpha = 2
    pw = [1]
    for _ in range(255):
        pw.append(mult(pw[-1], alpha))
    G = np.zeros((k, n), dtype=int)
    for row in range(1, k + 1):
        for col in range(1, n + 1):
            G[row - 1, col - 1] = pw[(row * col) % 255]
    # inverse of G(1:k,1:k) by Gauss-Jordan, then G = G_k_inv * G
    A = G[:, :k].copy()
    I = np.eye(k, dtype=int)
    for c in range(k):
        p = next(r for r in range(c, k) if A[r, c] != 0)
        if p != c:
            A[[c, p]] = A[[p, c]]
            I[[c, p]] = I[[p, c]]
        iv = inv(int(A[c, c]))
        A[c] = [mult(int(v), iv) for v in A[c]]
        I[c] = [mult(int(v), iv) for v in I[c]]
        for r in range(k):
            if r != c and A[r, c] != 0:
                f = int(A[r, c])
                A[r] = [add(int(A[r, t]), mult(f, int(A[c, t]))) for t in range(k)]
                I[r] = [add(int(I[r, t]), mult(f, int(I[c, t]))) for t in range(k)]
    out = np.zeros((k, n), dtype=int)
    for r in range(k):
        for c in range(n):
            s = 0
            for t in range(k):
                s = add(s, mult(int(I[r, t]), int(G[t, c])))
            out[r, c] = s
    return out


def rs_decode(recv_vec_ind, recv_vals, n, k, G):
    """Matlab/My_RS_Decode_Optimize_With_GFTables.m:15-118 (recv_vec_ind 1-based ascending)."""
    G1 = M1(G)
    GJ = M1(np.zeros((k, k)))
    for ii in range(1, k + 1):
        for t in range(1, k + 1):
            GJ[ii, t] = G1[t, recv_vec_ind[ii - 1]]
    num_sys_symbols = sum(1 for r in recv_vec_ind if r <= k)
    bit_order_vec = M1(np.arange(1, k + 1))
    for ii in range(1, num_sys_symbols + 1):
        col_ind, ind = 0, 1
        while col_ind == 0:
            if GJ[ii, ind] != 0:
                col_ind = ind
            ind += 1
        tmp = GJ.a[:, ii - 1].copy()
        GJ.a[:, ii - 1] = GJ.a[:, col_ind - 1]
        GJ.a[:, col_ind - 1] = tmp
        temp_col_ind = bit_order_vec[ii]
        bit_order_vec[ii] = col_ind
        bit_order_vec[col_ind] = temp_col_ind
    acc = M1(recv_vals)
    row_index = num_sys_symbols + 1
    swap_ind = row_index + 1
    NotDone = 1
    while row_index <= k and NotDone == 1:
        for jj in range(1, num_sys_symbols + 1):
            acc[row_index] = add(acc[row_index], mult(GJ[row_index, jj], acc[jj]))
            GJ[row_index, jj] = 0
        for jj in range(num_sys_symbols + 1, row_index):
            acc[row_index] = add(acc[row_index], mult(GJ[row_index, jj], acc[jj]))
            row_multiplier = GJ[row_index, jj]
            for ll in range(jj, k + 1):
                GJ[row_index, ll] = add(GJ[row_index, ll], mult(row_multiplier, GJ[jj, ll]))
        if GJ[row_index, row_index] != 0:
            GF_mult = inv(GJ[row_index, row_index])
            for ll in range(row_index, k + 1):
                GJ[row_index, ll] = mult(GF_mult, GJ[row_index, ll])
            acc[row_index] = mult(GF_mult, acc[row_index])
            row_index += 1
            swap_ind = row_index + 1
        else:
            if swap_ind > k:
                NotDone = 0
            else:
                tmp = GJ.a[row_index - 1, :].copy()
                GJ.a[row_index - 1, :] = GJ.a[swap_ind - 1, :]
                GJ.a[swap_ind - 1, :] = tmp
                acc[row_index], acc[swap_ind] = acc[swap_ind], acc[row_index]
                swap_ind += 1
    for ii in range(k - 1, num_sys_symbols, -1):
        for jj in range(ii + 1, k + 1):
            acc[ii] = add(acc[ii], mult(acc[jj], GJ[ii, jj]))
            GJ[ii, jj] = 0
    final_output = M1(np.zeros(k))
    for ii in range(1, num_sys_symbols + 1):
        final_output[bit_order_vec[ii]] = int(recv_vals[ii - 1])
    for ii in range(num_sys_symbols + 1, k + 1):
        final_output[bit_order_vec[ii]] = acc[ii]
    return final_output.a.copy()


def bursty_step(current_state, alpha, beta, good_transition_bias, rand_num, state_rand_num):
    """Matlab/Bursty_Error_Channel_Model_Generator.m:12-47."""
    transition = 0.1
    Prob_1_given_0 = transition / good_transition_bias
    Prob_0_given_1 = transition
    error_out = 0
    if current_state == 0:
        if rand_num <= alpha:
            error_out = 1
        next_state = 1 if state_rand_num <= Prob_1_given_0 else current_state
    else:
        if rand_num <= beta:
            error_out = 1
        next_state = 0 if state_rand_num <= Prob_0_given_1 else current_state
    return error_out, next_state


def fpga_perf_decoder(vlist, n_ldpc, k_ldpc, is_erasure, symbol, num_iter):
    """Second, independent restatement of OpenCL/device/ldpc_erasure_decoder_perf_tests.cl:56-236 (one frame), written
    with Python lists and 1-based Vlist rows [deg, cols...] exactly as the kernel indexes them.  `symbol` holds one
    integer per codeword symbol (it stands for the 128 64-bit words, which all go through the same XORs)."""
    codeword = [[symbol[i], int(is_erasure[i])] for i in range(n_ldpc)]
    codeword2 = [[symbol[i], int(is_erasure[i])] for i in range(n_ldpc)]
    num_parity_checks = n_ldpc - k_ldpc
    iter_ind, stop_sig = 0, 0
    while iter_ind < num_iter and stop_sig == 0:
        for cw, lo, hi in ((codeword, 0, num_parity_checks // 2), (codeword2, num_parity_checks // 2, num_parity_checks)):
            for k in range(lo, hi):
                acc, num_erasures, erasure_ind = 0, 0, 0
                for ii in range(vlist[k][0]):
                    j = vlist[k][ii + 1] - 1
                    acc ^= cw[j][0]
                    if cw[j][1] == 1:
                        num_erasures += 1
                        erasure_ind = j
                if num_erasures == 1:
                    cw[erasure_ind][1] = 0
                    cw[erasure_ind][0] = acc
        num_current_correct = 0
        for ii in range(n_ldpc):
            ssum = codeword[ii][1] + codeword2[ii][1]
            if ssum == 1:
                if codeword[ii][1]:
                    codeword[ii] = [codeword2[ii][0], 0]
                else:
                    codeword2[ii] = [codeword[ii][0], 0]
                num_current_correct += 1
            elif ssum == 0 and ii < k_ldpc:
                num_current_correct += 1
        if num_current_correct == k_ldpc:
            stop_sig = 1
        iter_ind += 1
    num_final_erasures = sum(codeword[ii][1] for ii in range(k_ldpc))
    return num_final_erasures, iter_ind, [c[1] for c in codeword], [c[0] for c in codeword]
