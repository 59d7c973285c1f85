"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/ldpc_erasure_amd.h declares,
its host GF tables equal the reference's .mat tables, and it fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from ldpc_erasure_codes_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    syms = set()
    for name in ("ldpc_erasure_amd.h", "ldpc_erasure_amd_wire.h", "ldpc_erasure_amd_multi.h"):   # every header under include/ that declares entry points
        txt = open(os.path.join(ROOT, "include", name)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        syms |= set(re.findall(r"\b(ldpc_amd_[a-z0-9_]+)\s*\(", txt))
    return sorted(syms)


def test_library_exports_every_declared_symbol():
    L = api.load_library()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"{s} declared in the header but not exported"
    assert sorted(api.EXPORTS) == syms


def test_host_gf_tables_equal_reference_mat():
    ref = np.load(os.path.join(ROOT, "tests", "golden", "gf256_tables_ref.npz"))
    mult, inv = api.gf_tables()
    assert np.array_equal(mult, ref["GF_mult_lookup"])
    assert np.array_equal(inv[1:], ref["GF_inv_lookup"])


def test_code_rom_matches_reference_params():
    assert api.code_params(0) == [2000, 1000, 0, 999, 250, 125]
    assert api.code_params(1) == [2040, 1530, 1000, 1509, 255, 192]
    assert api.code_params(2)[:2] == [4000, 2000]
    with pytest.raises(api.LdpcAmdError):
        api.code_params(99)


def test_symbol_type_layout_matches_reference():
    # OpenCL/host/src/main.cpp:44-47: 128 x unsigned long + 1 flag byte -> 1032 bytes with natural alignment
    class Sym(ctypes.Structure):
        _fields_ = [("symbol", ctypes.c_ulong * 128), ("is_erasure", ctypes.c_ubyte)]
    assert ctypes.sizeof(Sym) == 1032


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.LdpcAmdError):
        api.Context(0)
