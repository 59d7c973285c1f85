"""Code tables (host side).

The built-in code indices follow the reference's code ROM (`ldpc_params`,
OpenCL/device/LDPC_Vlist_data.h:10-14 / OpenCL/host/inc/Main_LDPC_header.h:10-14):

    0 -> (2000, 1000)   RS-equivalent (250, 125)
    1 -> (2040, 1530)   RS-equivalent (255, 192)

and are extended with the other matrices the reference ships or names:

    2 -> (4000, 2000)   Matlab/n4000_k2000_no6cycles_triangleForm.mat
    3 -> (4080, 3060)   named at Matlab/LDPCErasureCodes_MessagePassingAlgSim.m:42 but NOT shipped:
                        synthesised by tools/hgen, present only if the fixture exists.

The H structure is stored as CSR fixtures under data/ (exported from the reference's .mat DATA files by
tools/export_reference_data.py).  The GF(256) coefficients `H_sparse_nb` are not in the reference
(SURVEY.md section 7.2 "Missing inputs"); they are drawn from the committed seed with the reference's
rule "uniform on 1..255 per non-zero" (Matlab/ErasureCodes_NonBinaryLDPCSim.m:51-58).
"""
import os
import struct
from dataclasses import dataclass

import numpy as np

from . import synth

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# code_ind -> (fixture name, n, k, RS_n, RS_k)
BUILTIN = {
    0: ("n2000_k1000", 2000, 1000, 250, 125),
    1: ("n2040_k1530", 2040, 1530, 255, 192),
    2: ("n4000_k2000", 4000, 2000, 250, 125),
    3: ("n4080_k3060", 4080, 3060, 255, 192),
}

DEFAULT_COEF_SEED = {0: 2000, 1: 2040, 2: 4000, 3: 4080}


@dataclass
class Code:
    n: int
    k: int
    row_ptr: np.ndarray  # uint32 [m+1]
    cols: np.ndarray     # uint16 [nnz], 0-based, ascending per row
    coefs: np.ndarray    # uint8  [nnz], 1..255
    name: str = "custom"
    rs_n: int = 255
    rs_k: int = 0

    @property
    def m(self):
        return self.n - self.k

    @property
    def nnz(self):
        return int(self.row_ptr[-1])

    def binary(self):
        """Same structure with every coefficient 1 (the GF(2) siblings' H_sparse)."""
        return Code(self.n, self.k, self.row_ptr, self.cols, np.ones_like(self.coefs), self.name + "_bin",
                    self.rs_n, self.rs_k)

    def dense(self):
        H = np.zeros((self.m, self.n), dtype=np.uint8)
        for r in range(self.m):
            s, e = int(self.row_ptr[r]), int(self.row_ptr[r + 1])
            H[r, self.cols[s:e]] = self.coefs[s:e]
        return H


def read_csr(path):
    with open(path, "rb") as f:
        magic = f.read(8)
        if magic != b"LDPCCSR1":
            raise ValueError(f"{path}: bad magic {magic!r}")
        n, k, m, nnz = struct.unpack("<4I", f.read(16))
        row_ptr = np.frombuffer(f.read(4 * (m + 1)), dtype="<u4").copy()
        cols = np.frombuffer(f.read(2 * nnz), dtype="<u2").copy()
    if m != n - k or row_ptr[-1] != nnz or cols.size != nnz:
        raise ValueError(f"{path}: inconsistent header")
    return n, k, row_ptr, cols


def builtin_path(code_ind):
    return os.path.join(DATA_DIR, f"code_{BUILTIN[code_ind][0]}.csr.bin")


def have_builtin(code_ind):
    return code_ind in BUILTIN and os.path.exists(builtin_path(code_ind))


def load_builtin(code_ind, coef_seed=None, binary=False):
    name, n, k, rs_n, rs_k = BUILTIN[code_ind]
    n2, k2, row_ptr, cols = read_csr(builtin_path(code_ind))
    assert (n2, k2) == (n, k)
    if binary:
        coefs = np.ones(cols.size, dtype=np.uint8)
    else:
        seed = DEFAULT_COEF_SEED[code_ind] if coef_seed is None else coef_seed
        coefs = synth.coefs(seed, cols.size)
    return Code(n, k, row_ptr, cols, coefs, name, rs_n, rs_k)


def from_dense(H, k):
    """Small hand-written codes for tests: H is an (n-k) x n array of GF(256) coefficients."""
    H = np.asarray(H, dtype=np.uint8)
    m, n = H.shape
    assert m == n - k
    row_ptr = [0]
    cols, coefs = [], []
    for r in range(m):
        nz = np.nonzero(H[r])[0]
        cols.extend(nz.tolist())
        coefs.extend(H[r, nz].tolist())
        row_ptr.append(len(cols))
    return Code(n, k, np.array(row_ptr, dtype=np.uint32), np.array(cols, dtype=np.uint16),
                np.array(coefs, dtype=np.uint8))
