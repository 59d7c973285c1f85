"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

May be imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg (see
oracle/oracle.h).  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class GF(C.Structure):
    _fields_ = [("add", (C.c_uint8 * 256) * 256), ("mult", (C.c_uint8 * 256) * 256),
                ("inv", C.c_uint8 * 255), ("antilog", C.c_uint8 * 256), ("log", C.c_int16 * 256)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    L.oracle_gf_default.restype = C.POINTER(GF)
    L.oracle_gf_build.argtypes = [C.POINTER(GF), C.c_int]
    L.oracle_code_create.restype = C.c_void_p
    L.oracle_code_create.argtypes = [C.c_int, C.c_int, u32p, u16p, u8p]
    L.oracle_code_destroy.argtypes = [C.c_void_p]
    L.oracle_ldpc_encode.argtypes = [C.c_void_p, u8p, u8p]
    L.oracle_ldpc_encode_packets.argtypes = [C.c_void_p, C.c_int, u8p, u8p]
    L.oracle_ldpc_hybridml_nonbinary_decode.argtypes = [C.c_void_p, i16p, C.c_int, C.c_int, i16p,
                                                        C.POINTER(C.c_int), i32p]
    L.oracle_ldpc_hybridml_nonbinary_decode_packets.argtypes = [C.c_void_p, C.c_int, u8p, u8p, C.c_int, C.c_int,
                                                                u8p, u8p, C.POINTER(C.c_int), i32p]
    L.oracle_ldpc_binary_mp_decode.argtypes = [C.c_void_p, i16p, C.c_int, i16p, C.POINTER(C.c_int)]
    L.oracle_ldpc_binary_hybridml_decode.argtypes = [C.c_void_p, i16p, C.c_int, i16p, C.POINTER(C.c_int), i32p]
    L.oracle_rs_generator.argtypes = [C.c_int, C.c_int, u8p]
    L.oracle_rs_encode.argtypes = [C.c_int, C.c_int, u8p, u8p, u8p]
    L.oracle_rs_decode.argtypes = [C.c_int, C.c_int, u8p, u16p, u8p, u8p]
    L.oracle_bursty_channel_step.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                             C.POINTER(C.c_int)]
    L.oracle_synth_coefs.argtypes = [C.c_uint64, C.c_int, u8p]
    L.oracle_synth_source.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_int, u8p]
    L.oracle_synth_erasures_uniform.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_double, u8p]
    L.oracle_synth_erasures_bursty.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double,
                                               C.c_double, u8p]
    L.oracle_fpga_data_in_erasures.argtypes = [C.c_int, C.c_int, C.c_int64, u8p]
    L.oracle_fpga_perf_decoder_frame.argtypes = [C.c_void_p, C.c_int, u8p, C.c_void_p, C.POINTER(C.c_int)]
    L.oracle_threefry4x32_20.argtypes = [u32p, u32p, u32p]
    L.oracle_ldpc_decode_batch_s1.argtypes = [C.c_void_p, C.c_int, u8p, u8p, C.c_int, C.c_int, u8p, i32p, i32p, i32p]
    _LIB = L
    return L


def gf_tables(prim_poly=None):
    """dict of numpy arrays add[256,256], mult[256,256], inv[255], antilog[256], log[256]."""
    L = lib()
    if prim_poly is None:
        t = L.oracle_gf_default().contents
    else:
        t = GF()
        L.oracle_gf_build(C.byref(t), prim_poly)
    return {
        "add": np.ctypeslib.as_array(t.add).reshape(256, 256).copy(),
        "mult": np.ctypeslib.as_array(t.mult).reshape(256, 256).copy(),
        "inv": np.ctypeslib.as_array(t.inv).copy(),
        "antilog": np.ctypeslib.as_array(t.antilog).copy(),
        "log": np.ctypeslib.as_array(t.log).copy(),
    }


class OracleCode:
    """Owns an oracle_code built from an ldpc_erasure_codes_amd.codes.Code-like object."""

    def __init__(self, code):
        self.n, self.k, self.m = code.n, code.k, code.n - code.k
        self._h = lib().oracle_code_create(code.n, code.k, np.ascontiguousarray(code.row_ptr, dtype=np.uint32),
                                           np.ascontiguousarray(code.cols, dtype=np.uint16),
                                           np.ascontiguousarray(code.coefs, dtype=np.uint8))
        if not self._h:
            raise ValueError("oracle_code_create rejected the code")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_code_destroy(self._h)
            self._h = None

    def encode(self, source):
        source = np.ascontiguousarray(source, dtype=np.uint8)
        if source.ndim == 1:
            cw = np.zeros(self.n, dtype=np.uint8)
            lib().oracle_ldpc_encode(self._h, source, cw)
            return cw
        # [k, S] packets: encode every lane
        k, S = source.shape
        cw = np.zeros((self.n, S), dtype=np.uint8)
        lib().oracle_ldpc_encode_packets(self._h, S, source, cw)
        return cw

    def decode(self, recv, itenum=10, do_ml=1):
        """Scalar frame, Matlab conventions: recv int16[n] with -1 = erasure. -> (msg, iterations, info, rc)"""
        recv = np.ascontiguousarray(recv, dtype=np.int16)
        msg = np.zeros(self.n, dtype=np.int16)
        it = C.c_int(0)
        info = np.zeros(3, dtype=np.int32)
        rc = lib().oracle_ldpc_hybridml_nonbinary_decode(self._h, recv, itenum, do_ml, msg, C.byref(it), info)
        return msg, it.value, info, rc

    def decode_packets(self, sym, erased, itenum=10, do_ml=1):
        """sym uint8[n,S], erased uint8[n] -> (out[n,S], out_erased[n], iterations, info, rc)"""
        sym = np.ascontiguousarray(sym, dtype=np.uint8)
        erased = np.ascontiguousarray(erased, dtype=np.uint8)
        n, S = sym.shape
        out = np.zeros((n, S), dtype=np.uint8)
        oe = np.zeros(n, dtype=np.uint8)
        it = C.c_int(0)
        info = np.zeros(3, dtype=np.int32)
        rc = lib().oracle_ldpc_hybridml_nonbinary_decode_packets(self._h, S, sym, erased, itenum, do_ml, out, oe,
                                                                 C.byref(it), info)
        return out, oe, it.value, info, rc

    def decode_batch_s1(self, sym, erased, itenum=10, do_ml=1):
        sym = np.ascontiguousarray(sym, dtype=np.uint8)
        erased = np.ascontiguousarray(erased, dtype=np.uint8)
        nframes = sym.shape[0]
        out = np.zeros_like(sym)
        sweeps = np.zeros(nframes, dtype=np.int32)
        residual = np.zeros(nframes, dtype=np.int32)
        status = np.zeros(nframes, dtype=np.int32)
        lib().oracle_ldpc_decode_batch_s1(self._h, nframes, sym, erased, itenum, do_ml, out, sweeps, residual, status)
        return out, sweeps, residual, status

    def binary_mp(self, recv, itenum=50):
        recv = np.ascontiguousarray(recv, dtype=np.int16)
        msg = np.zeros(self.n, dtype=np.int16)
        it = C.c_int(0)
        lib().oracle_ldpc_binary_mp_decode(self._h, recv, itenum, msg, C.byref(it))
        return msg, it.value

    def fpga_perf_decoder(self, erased, num_iter, payload=None):
        """OpenCL/device/ldpc_erasure_decoder_perf_tests.cl frame loop -> (systematic erasures left, iterations,
        final is_erasure flags of the first copy, final payload words or None)."""
        er = np.ascontiguousarray(erased, dtype=np.uint8).copy()
        pl = None if payload is None else np.ascontiguousarray(payload, dtype=np.uint64).copy()
        it = C.c_int(0)
        left = lib().oracle_fpga_perf_decoder_frame(self._h, num_iter, er, None if pl is None else pl.ctypes.data, C.byref(it))
        return left, it.value, er, pl

    def binary_hybrid(self, recv, itenum=10):
        recv = np.ascontiguousarray(recv, dtype=np.int16)
        msg = np.zeros(self.n, dtype=np.int16)
        it = C.c_int(0)
        info = np.zeros(3, dtype=np.int32)
        rc = lib().oracle_ldpc_binary_hybridml_decode(self._h, recv, itenum, msg, C.byref(it), info)
        return msg, it.value, info, rc


def rs_generator(n, k):
    g = np.zeros((k, n), dtype=np.uint8)
    if lib().oracle_rs_generator(n, k, g) != 0:
        raise ValueError("singular Vandermonde block")
    return g


def rs_encode(g, source):
    k, n = g.shape
    cw = np.zeros(n, dtype=np.uint8)
    lib().oracle_rs_encode(n, k, np.ascontiguousarray(g), np.ascontiguousarray(source, dtype=np.uint8), cw)
    return cw


def rs_decode(g, recv_ind, recv_val):
    k, n = g.shape
    msg = np.zeros(k, dtype=np.uint8)
    rc = lib().oracle_rs_decode(n, k, np.ascontiguousarray(g), np.ascontiguousarray(recv_ind, dtype=np.uint16),
                                np.ascontiguousarray(recv_val, dtype=np.uint8), msg)
    return msg, rc


def synth_coefs(seed, nnz):
    out = np.zeros(nnz, dtype=np.uint8)
    lib().oracle_synth_coefs(seed, nnz, out)
    return out


def synth_source(seed, frame0, nframes, k, S):
    out = np.zeros((nframes, k, S), dtype=np.uint8)
    lib().oracle_synth_source(seed, frame0, nframes, k, S, out)
    return out


def synth_erasures_uniform(seed, frame0, nframes, n, per):
    out = np.zeros((nframes, n), dtype=np.uint8)
    lib().oracle_synth_erasures_uniform(seed, frame0, nframes, n, per, out)
    return out


def synth_erasures_bursty(seed, frame0, nframes, n, alpha, beta, bias):
    out = np.zeros((nframes, n), dtype=np.uint8)
    lib().oracle_synth_erasures_bursty(seed, frame0, nframes, n, alpha, beta, bias, out)
    return out


def fpga_data_in_erasures(seed, per64, nframes, n):
    out = np.zeros((nframes, n), dtype=np.uint8)
    lib().oracle_fpga_data_in_erasures(seed, per64, nframes * n, out)
    return out


def threefry4x32_20(ctr, key):
    out = np.zeros(4, dtype=np.uint32)
    lib().oracle_threefry4x32_20(np.ascontiguousarray(ctr, dtype=np.uint32), np.ascontiguousarray(key, dtype=np.uint32), out)
    return out
