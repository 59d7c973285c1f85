#!/usr/bin/env python3
"""Export the reference's DATA files into small committed fixtures.

Runs only in the build container (needs /root/reference).  Nothing here copies reference
source text: it reads

  * Matlab/*.mat  (MAT-v5, loaded with scipy.io.loadmat -- no code is executed from the files)
      - the three binary parity-check matrices H_sparse            
      - GF_256_add_mult_inv_tables.mat (add/mult/inv lookup tables)
  * OpenCL/device/LDPC_Vlist_data.h -- parsed as a table of integers (the "code ROM",
    SURVEY.md section 8 a7); only the numbers are kept.

and writes

  ldpc_erasure_codes_amd/data/code_<name>.csr.bin   CSR of each H (0-based, ascending columns)
  tests/golden/gf256_tables_ref.npz                  the reference's GF tables (KAT for the oracle)
  tests/golden/code_rom_ref.npz                      ldpc_params + Vlist_master rows (KAT for code tables)

CSR file layout (little endian):
  char[8] magic "LDPCCSR1"; u32 n, k, m, nnz; u32 row_ptr[m+1]; u16 cols[nnz]
"""
import os
import re
import struct
import sys

import numpy as np
import scipy.io as sio

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "ldpc_erasure_codes_amd", "data")
GOLD = os.path.join(ROOT, "tests", "golden")

CODES = {
    # fixture name -> .mat file (variable H_sparse)
    "n2040_k1530": "n2040_k1530_irreg_H_no6cycles_triangleForm.mat",
    "n4000_k2000": "n4000_k2000_no6cycles_triangleForm.mat",
    "n2000_k1000": "n2000_k1000_no6cycles_triangleForm_OpenCL_H.mat",
}


def write_csr(path, n, k, row_ptr, cols):
    m = n - k
    with open(path, "wb") as f:
        f.write(b"LDPCCSR1")
        f.write(struct.pack("<4I", n, k, m, len(cols)))
        f.write(np.asarray(row_ptr, dtype="<u4").tobytes())
        f.write(np.asarray(cols, dtype="<u2").tobytes())


def export_codes():
    for name, fn in CODES.items():
        H = sio.loadmat(os.path.join(REF, "Matlab", fn))["H_sparse"].tocsr()
        H.sort_indices()
        m, n = H.shape
        assert np.all(H.data == 1.0), "reference H is binary"
        k = n - m
        # triangle form: last non-zero of row i is column k+i (SURVEY.md section 8 preamble)
        last = H.indices[H.indptr[1:] - 1]
        assert np.array_equal(last, k + np.arange(m)), name
        write_csr(os.path.join(DATA, f"code_{name}.csr.bin"), n, k, H.indptr, H.indices)
        print(f"{name}: n={n} k={k} m={m} nnz={H.nnz}")


def export_gf_tables():
    d = sio.loadmat(os.path.join(REF, "Matlab", "GF_256_add_mult_inv_tables.mat"))
    np.savez_compressed(
        os.path.join(GOLD, "gf256_tables_ref.npz"),
        GF_add_lookup=d["GF_add_lookup"].astype(np.uint8),
        GF_mult_lookup=d["GF_mult_lookup"].astype(np.uint8),
        GF_inv_lookup=d["GF_inv_lookup"].astype(np.uint8).reshape(-1),
    )
    print("gf tables:", {k: v.shape for k, v in d.items() if not k.startswith("__")})


def export_code_rom():
    txt = open(os.path.join(REF, "OpenCL", "device", "LDPC_Vlist_data.h")).read()
    txt = re.sub(r"//[^\n]*", "", txt)
    m = re.search(r"ldpc_params\[2\]\[6\]\s*=\s*\{(.*?)\};", txt, re.S)
    params = np.array([int(x) for x in re.findall(r"-?\d+", m.group(1))], dtype=np.int32).reshape(2, 6)
    m = re.search(r"parity_check_mat_Vlist_master\[(\d+)\]\[(\d+)\]\s*=\s*\{(.*)\};", txt, re.S)
    rows, width = int(m.group(1)), int(m.group(2))
    vals = np.array([int(x) for x in re.findall(r"-?\d+", m.group(3))], dtype=np.int16)
    assert vals.size == rows * width, (vals.size, rows, width)
    np.savez_compressed(os.path.join(GOLD, "code_rom_ref.npz"), ldpc_params=params,
                        vlist_master=vals.reshape(rows, width))
    print("code rom:", params.tolist(), (rows, width))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    os.makedirs(DATA, exist_ok=True)
    os.makedirs(GOLD, exist_ok=True)
    export_codes()
    export_gf_tables()
    export_code_rom()
