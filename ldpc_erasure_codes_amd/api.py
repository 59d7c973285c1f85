"""ctypes binding of the C ABI (include/ldpc_erasure_amd.h -> libldpc_erasure_amd.so).

Host-side mirror of the reference's interfaces for the hot path, so that tests read like the reference's
own harnesses:

    Context()                        ~ init_opencl()            OpenCL/host/src/main.cpp:439
    Context.close()                  ~ cleanup()                main.cpp:668
    Context.decode(...)              ~ My_LDPC_HybridML_NonBinary_Erasure_Decoder(recv, Vlist, Clist, H, n, k, ...)
                                       Matlab/My_LDPC_HybridML_NonBinary_Erasure_Decoder.m:4   (batched)
    Context.rs_decode(...)           ~ My_RS_Decode(recv_vec_ind, recv_vec_gf256_val, m, n, k, ...)
                                       Matlab/My_RS_Decode.m:14                                   (batched)

There is NO CPU fallback here: if the HIP library is missing or no gfx950 device is present every call
raises.  Arrays may be numpy (host pointers; the library stages them) or torch CUDA tensors (device
pointers; asynchronous on the context's stream).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libldpc_erasure_amd.so")

OK = 0
DEVICE_PTRS = 1
INPLACE = 2
ST_MP_DONE, ST_ML_SOLVED, ST_ML_RANKDEF, ST_ML_SKIPPED = 0, 1, 2, 3
# LDPC_AMD_PROF_*: "apply" = both tiers of the packet kernel ("apply_tier2" is the tier-2 launch alone), "ml" = factorisation + solve
# ("ml_solve" is the solve kernel alone)
PROF_KINDS = ("peel", "apply", "ml", "apply_tier2", "ml_solve")

# every symbol include/ldpc_erasure_amd.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "ldpc_amd_init", "ldpc_amd_cleanup", "ldpc_amd_last_error", "ldpc_amd_set_stream", "ldpc_amd_synchronize", "ldpc_amd_configure",
    "ldpc_amd_code_params", "ldpc_amd_load_builtin_code", "ldpc_amd_register_code", "ldpc_amd_code_info", "ldpc_amd_encode_info", "ldpc_amd_knobs",
    "ldpc_amd_code_csr", "ldpc_amd_decode_batch", "ldpc_amd_encode_batch", "ldpc_amd_rs_create",
    "ldpc_amd_rs_generator", "ldpc_amd_rs_encode_batch", "ldpc_amd_rs_decode_batch", "ldpc_amd_rs_bad_blocks", "ldpc_amd_synth_source",
    "ldpc_amd_synth_erasures_uniform", "ldpc_amd_synth_erasures_bursty", "ldpc_amd_data_in", "ldpc_amd_data_in_at",
    "ldpc_amd_shard_frames", "ldpc_amd_group_create", "ldpc_amd_group_destroy", "ldpc_amd_group_size", "ldpc_amd_group_device", "ldpc_amd_group_ctx",
    "ldpc_amd_group_last_error", "ldpc_amd_group_load_builtin_code", "ldpc_amd_group_register_code", "ldpc_amd_group_decode_batch",
    "ldpc_amd_group_decode_resident", "ldpc_amd_group_fpga_run", "ldpc_amd_group_bench_resident", "ldpc_amd_ldpc_erasure_decoder", "ldpc_amd_data_out",
    "ldpc_amd_ldpc_erasure_decoder_perf_tests", "ldpc_amd_fpga_frame_stats", "ldpc_amd_profile_kernel_name", "ldpc_amd_last_plan", "ldpc_amd_ml_stats",
    "ldpc_amd_fec_header_pack", "ldpc_amd_fec_header_unpack", "ldpc_amd_fec_packetize", "ldpc_amd_fec_rx_create",
    "ldpc_amd_fec_rx_destroy", "ldpc_amd_fec_rx_push", "ldpc_amd_fec_rx_push_many", "ldpc_amd_fec_rx_flush", "ldpc_amd_fec_rx_dropped",
    "ldpc_amd_set_profiling", "ldpc_amd_get_profile", "ldpc_amd_selftest", "ldpc_amd_copy_probe", "ldpc_amd_gf_tables", "ldpc_amd_version",
]


class LdpcAmdError(RuntimeError):
    pass


class ErrorType(C.Structure):
    _fields_ = [("num_LDPC_errors", C.c_int), ("num_RS_errors", C.c_int)]


_lib = None


def load_library():
    """Loads libldpc_erasure_amd.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LdpcAmdError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
    # PyTorch-ROCm ships its own libamdhip64.so.7 / libhsa-runtime64; two HIP runtimes in one process do not
    # both see the GPU.  Loading torch first makes our library bind to the runtime torch uses (same SONAME),
    # so device pointers, streams and events are shared between the two.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional plumbing: without it the system ROCm runtime is used
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64
    L.ldpc_amd_init.argtypes = [i32, C.POINTER(vp)]
    L.ldpc_amd_cleanup.argtypes = [vp]
    L.ldpc_amd_cleanup.restype = None
    L.ldpc_amd_last_error.argtypes = [vp]
    L.ldpc_amd_last_error.restype = C.c_char_p
    L.ldpc_amd_set_stream.argtypes = [vp, vp]
    L.ldpc_amd_synchronize.argtypes = [vp]
    if hasattr(L, "ldpc_amd_configure"):   # (tools/ab_lib.py and tools/time_latency.py also load older builds of the library)
        L.ldpc_amd_configure.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.ldpc_amd_code_params.argtypes = [i32, C.POINTER(i32)]
    L.ldpc_amd_load_builtin_code.argtypes = [vp, i32, u64]
    L.ldpc_amd_register_code.argtypes = [vp, i32, i32, vp, vp, vp]
    L.ldpc_amd_code_info.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.ldpc_amd_encode_info.argtypes = [vp, i32, C.POINTER(i32)]
    if hasattr(L, "ldpc_amd_knobs"):
        L.ldpc_amd_knobs.argtypes = [vp, C.c_char_p, i32]
    # multi-device layer (include/ldpc_erasure_amd_multi.h)
    L.ldpc_amd_shard_frames.argtypes = [i64, i32, i32, C.POINTER(i64), C.POINTER(i64)]
    L.ldpc_amd_shard_frames.restype = None
    L.ldpc_amd_group_create.argtypes = [i32, C.POINTER(i32), C.POINTER(vp)]
    L.ldpc_amd_group_destroy.argtypes = [vp]
    L.ldpc_amd_group_destroy.restype = None
    L.ldpc_amd_group_size.argtypes = [vp]
    L.ldpc_amd_group_device.argtypes = [vp, i32]
    L.ldpc_amd_group_ctx.argtypes = [vp, i32]
    L.ldpc_amd_group_ctx.restype = vp
    L.ldpc_amd_group_last_error.argtypes = [vp]
    L.ldpc_amd_group_last_error.restype = C.c_char_p
    L.ldpc_amd_group_load_builtin_code.argtypes = [vp, i32, u64]
    L.ldpc_amd_group_register_code.argtypes = [vp, i32, i32, vp, vp, vp]
    L.ldpc_amd_group_decode_batch.argtypes = [vp, i32, i32, i64, vp, vp, i32, i32, vp, vp, vp, vp]
    L.ldpc_amd_group_decode_resident.argtypes = [vp, i32, i32, i64, C.POINTER(vp), C.POINTER(vp), i32, i32, C.POINTER(vp), C.POINTER(vp), vp, vp,
                                                 C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ldpc_amd_group_fpga_run.argtypes = [vp, C.c_ushort, i32, i32, i32, C.c_long, C.c_short, i32, C.POINTER(ErrorType)]
    L.ldpc_amd_group_bench_resident.argtypes = [vp, i32, u64, i32, i64, C.c_double, i32, i32, C.POINTER(C.c_double)]
    L.ldpc_amd_data_in_at.argtypes = [vp, vp, C.c_ushort, i32, i32, i32, C.c_long, C.c_long]
    L.ldpc_amd_code_csr.argtypes = [vp, i32, vp, vp, vp]
    L.ldpc_amd_decode_batch.argtypes = [vp, i32, i32, i64, vp, vp, i32, i32, vp, vp, vp, vp, C.c_uint]
    L.ldpc_amd_encode_batch.argtypes = [vp, i32, i32, i64, vp, vp, C.c_uint]
    L.ldpc_amd_rs_create.argtypes = [vp, i32, i32]
    L.ldpc_amd_rs_generator.argtypes = [vp, i32, vp]
    L.ldpc_amd_rs_encode_batch.argtypes = [vp, i32, i32, i64, vp, vp, C.c_uint]
    L.ldpc_amd_rs_decode_batch.argtypes = [vp, i32, i32, i64, vp, vp, vp, C.c_uint]
    if hasattr(L, "ldpc_amd_rs_bad_blocks"):
        L.ldpc_amd_rs_bad_blocks.argtypes = [vp, C.POINTER(C.c_longlong)]
    L.ldpc_amd_synth_source.argtypes = [vp, u64, i64, i64, i32, i32, vp]
    L.ldpc_amd_synth_erasures_uniform.argtypes = [vp, u64, i64, i64, i32, C.c_double, vp]
    L.ldpc_amd_synth_erasures_bursty.argtypes = [vp, u64, i64, i64, i32, C.c_double, C.c_double, C.c_double, vp]
    L.ldpc_amd_data_in.argtypes = [vp, vp, C.c_ushort, i32, i32, i32, C.c_long]
    L.ldpc_amd_ldpc_erasure_decoder.argtypes = [vp, C.c_short, i32]
    L.ldpc_amd_ldpc_erasure_decoder_perf_tests.argtypes = [vp, C.c_short, i32]
    L.ldpc_amd_fpga_frame_stats.argtypes = [vp, C.c_long, vp, vp]
    # host-side wire format (include/ldpc_erasure_amd_wire.h)
    L.ldpc_amd_fec_header_pack.argtypes = [C.c_uint, C.c_uint, C.c_uint]
    L.ldpc_amd_fec_header_pack.restype = C.c_uint64
    L.ldpc_amd_fec_header_unpack.argtypes = [C.c_uint64, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    L.ldpc_amd_fec_header_unpack.restype = None
    L.ldpc_amd_fec_packetize.argtypes = [vp, C.c_long, i32, i32, C.c_uint, C.c_uint, vp]
    L.ldpc_amd_fec_rx_create.argtypes = [i32, i32, i32, C.POINTER(vp)]
    L.ldpc_amd_fec_rx_destroy.argtypes = [vp]
    L.ldpc_amd_fec_rx_destroy.restype = None
    L.ldpc_amd_fec_rx_push.argtypes = [vp, vp, vp, vp, C.POINTER(i32)]
    L.ldpc_amd_fec_rx_flush.argtypes = [vp, vp, vp, C.POINTER(i32)]
    L.ldpc_amd_fec_rx_push_many.argtypes = [vp, vp, C.c_long, vp, vp, vp, i32, C.POINTER(C.c_long)]
    L.ldpc_amd_fec_rx_dropped.argtypes = [vp]
    L.ldpc_amd_fec_rx_dropped.restype = C.c_long
    L.ldpc_amd_data_out.argtypes = [vp, vp, i32, C.c_long, C.POINTER(ErrorType)]
    L.ldpc_amd_set_profiling.argtypes = [vp, i32]
    L.ldpc_amd_get_profile.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i64)]
    L.ldpc_amd_last_plan.argtypes = [vp, C.POINTER(i32)]
    if hasattr(L, "ldpc_amd_ml_stats"):   # (absent from the older builds tools/ab_lib.py loads)
        L.ldpc_amd_ml_stats.argtypes = [vp, C.POINTER(C.c_longlong)]
    L.ldpc_amd_profile_kernel_name.argtypes = [vp, i32]
    L.ldpc_amd_profile_kernel_name.restype = C.c_char_p
    L.ldpc_amd_selftest.argtypes = [vp]
    L.ldpc_amd_copy_probe.argtypes = [vp, vp, vp, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
    L.ldpc_amd_gf_tables.argtypes = [vp, vp]
    L.ldpc_amd_version.restype = C.c_char_p
    _lib = L
    return L


def gf_tables():
    """(mult[256,256], inv[256]) the kernels use -- host computation, no GPU needed."""
    L = load_library()
    mult = np.zeros((256, 256), dtype=np.uint8)
    inv = np.zeros(256, dtype=np.uint8)
    L.ldpc_amd_gf_tables(mult.ctypes.data, inv.ctypes.data)
    return mult, inv


def code_params(code_ind):
    L = load_library()
    p = (C.c_int * 6)()
    if L.ldpc_amd_code_params(code_ind, p) != OK:
        raise LdpcAmdError(f"no built-in code {code_ind}")
    return list(p)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x):
    if x is None:
        return None
    if _is_torch(x):
        assert x.is_contiguous()
        return x.data_ptr()
    assert x.flags["C_CONTIGUOUS"]
    return x.ctypes.data


class Context:
    def __init__(self, device=0):
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.ldpc_amd_init(device, C.byref(h))
        if rc != OK:
            raise LdpcAmdError(f"ldpc_amd_init({device}) = {rc}: {self._L.ldpc_amd_last_error(None).decode()}")
        self._h = h
        self.device = device

    # -- life-cycle
    def close(self):
        if getattr(self, "_h", None):
            self._L.ldpc_amd_cleanup(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise LdpcAmdError(f"{what} = {rc}: {self._L.ldpc_amd_last_error(self._h).decode()}")
        return rc

    def set_stream(self, stream_handle):
        self._check(self._L.ldpc_amd_set_stream(self._h, C.c_void_p(stream_handle)), "set_stream")

    def synchronize(self):
        self._check(self._L.ldpc_amd_synchronize(self._h), "synchronize")

    def configure(self, key, value=None):
        """Sets a tuning / diagnostic knob of this context ("SCATTER_B", "LDPC_AMD_ML_SOLVE", ...); value None restores the
        default.  The LDPC_AMD_* environment variables are only the initial values, read once when the context is created."""
        v = None if value is None else str(value).encode()
        self._check(self._L.ldpc_amd_configure(self._h, key.encode(), v), "configure")

    def knobs(self):
        """The knobs of this context that are NOT at their shipped default, "NAME=value NAME=value" ('' = all defaults)."""
        buf = C.create_string_buffer(1024)
        self._check(self._L.ldpc_amd_knobs(self._h, buf, 1024), "knobs")
        return buf.value.decode()

    def configure_many(self, knobs):
        """{key: value or None} -> configure() for each."""
        for k, v in knobs.items():
            self.configure(k, v)

    def set_profiling(self, enable):
        """False / 0: off; True / 1: one bracket per kind of a call; 2: + the nested tier-2 and solve-kernel brackets."""
        self._check(self._L.ldpc_amd_set_profiling(self._h, int(enable)), "set_profiling")

    def get_profile(self):
        """{'peel': (ms, launches), 'apply': ..., 'ml': ..., 'apply_tier2': ..., 'ml_solve': ...} since the last call (synchronises)."""
        ms = (C.c_double * len(PROF_KINDS))()
        cnt = (C.c_int64 * len(PROF_KINDS))()
        self._check(self._L.ldpc_amd_get_profile(self._h, ms, cnt), "get_profile")
        return {name: (ms[i], cnt[i]) for i, name in enumerate(PROF_KINDS)}

    def profile_kernel_names(self):
        """{'peel': name, 'apply': name, 'ml': name}: the kernel instantiation the last launch of each kind used."""
        return {name: self._L.ldpc_amd_profile_kernel_name(self._h, i).decode() for i, name in enumerate(PROF_KINDS)}

    def last_plan(self):
        info = (C.c_int * 8)()
        self._check(self._L.ldpc_amd_last_plan(self._h, info), "last_plan")
        keys = ("frames_per_workgroup", "frames_per_cu", "tables_in_global", "peel_lds_bytes", "peel_lds_per_frame",
                "packet_bytes_per_workgroup", "tier1_cap", "two_tiers")
        return dict(zip(keys, list(info)))

    def ml_stats(self):
        """ML stage of the last decode: residual frames, frames solved through the fast path, frames its consistency test flagged
        (redone exactly), frames whose schedule did not fit the arena.  Synchronises."""
        st = (C.c_longlong * 4)()
        self._check(self._L.ldpc_amd_ml_stats(self._h, st), "ml_stats")
        return dict(zip(("residual_frames", "fast_path_frames", "flagged_frames", "deferred_frames"), [int(x) for x in st]))

    def copy_probe(self, src, dst, reps=10, nbytes=None):
        """Best single-launch device time (ms) of a streaming copy src -> dst (torch CUDA uint8 tensors of equal size;
        nbytes: copy only that many leading bytes), over `reps` launches of each of the probe's launch shapes."""
        ms = C.c_double(0.0)
        nbytes = min(nbytes or (1 << 62), src.numel() * src.element_size())
        self._check(self._L.ldpc_amd_copy_probe(self._h, src.data_ptr(), dst.data_ptr(), nbytes & ~15, reps, C.byref(ms)), "copy_probe")
        return ms.value

    def selftest(self):
        self._check(self._L.ldpc_amd_selftest(self._h), "selftest")

    # -- codes
    def load_builtin_code(self, code_ind, coef_seed):
        return self._check(self._L.ldpc_amd_load_builtin_code(self._h, code_ind, coef_seed), "load_builtin_code")

    def register_code(self, code):
        rp = np.ascontiguousarray(code.row_ptr, dtype=np.uint32)
        cols = np.ascontiguousarray(code.cols, dtype=np.uint16)
        coefs = np.ascontiguousarray(code.coefs, dtype=np.uint8)
        return self._check(self._L.ldpc_amd_register_code(self._h, code.n, code.k, rp.ctypes.data, cols.ctypes.data,
                                                          coefs.ctypes.data), "register_code")

    def code_info(self, code):
        n, k, nnz = C.c_int(), C.c_int(), C.c_int()
        self._check(self._L.ldpc_amd_code_info(self._h, code, C.byref(n), C.byref(k), C.byref(nnz)), "code_info")
        return n.value, k.value, nnz.value

    def code_csr(self, code):
        n, k, nnz = self.code_info(code)
        rp = np.zeros(n - k + 1, dtype=np.uint32)
        cols = np.zeros(nnz, dtype=np.uint16)
        coefs = np.zeros(nnz, dtype=np.uint8)
        self._check(self._L.ldpc_amd_code_csr(self._h, code, rp.ctypes.data, cols.ctypes.data, coefs.ctypes.data), "code_csr")
        return rp, cols, coefs

    # -- hot path
    def decode(self, code, sym, erased, max_sweeps=10, do_ml=1, out=None, sweeps=None, residual=None, status=None,
               inplace=False):
        """sym [F,n,S] (or [F,n] for S=1) uint8, erased [F,n] uint8.
        numpy in -> numpy out (synchronous).  torch CUDA tensors in -> torch out (asynchronous).
        Returns (out, sweeps, residual, status)."""
        n, k, _ = self.code_info(code)
        dev = _is_torch(sym)
        F = sym.shape[0]
        S = 1 if sym.ndim == 2 else sym.shape[2]
        assert sym.shape[1] == n and tuple(erased.shape) == (F, n)
        if dev:
            import torch
            mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=sym.device)  # noqa: E731
            if inplace:
                out = sym
            out = mk(tuple(sym.shape), torch.uint8) if out is None else out
            sweeps = mk((F,), torch.int32) if sweeps is None else sweeps
            residual = mk((F,), torch.int32) if residual is None else residual
            status = mk((F,), torch.int32) if status is None else status
        else:
            sym = np.ascontiguousarray(sym, dtype=np.uint8)
            erased = np.ascontiguousarray(erased, dtype=np.uint8)
            out = np.empty_like(sym) if out is None else out
            sweeps = np.empty(F, dtype=np.int32) if sweeps is None else sweeps
            residual = np.empty(F, dtype=np.int32) if residual is None else residual
            status = np.empty(F, dtype=np.int32) if status is None else status
        self._check(self._L.ldpc_amd_decode_batch(self._h, code, S, F, _ptr(sym), _ptr(erased), max_sweeps, do_ml,
                                                  _ptr(out), _ptr(sweeps), _ptr(residual), _ptr(status),
                                                  (DEVICE_PTRS if dev else 0) | (INPLACE if inplace else 0)), "decode_batch")
        return out, sweeps, residual, status

    def encode_info(self, code):
        """Static schedules of the code's systematic encoder: levels of the parity triangle, groups of the level-collapsed
        schedule (0: none), accumulators pulled, scatter entries left, longest pull list, and whether the last packet-mode
        encode of this context ran the grouped schedule."""
        info = (C.c_int * 6)()
        self._check(self._L.ldpc_amd_encode_info(self._h, code, info), "encode_info")
        return dict(zip(("levels", "groups", "pull_entries", "scatter_entries", "max_pull", "last_encode_grouped"), list(info)))

    def encode(self, code, source, out=None):
        """source [F,k,S] (or [F,k]) -> codeword [F,n,S] (or [F,n])."""
        n, k, _ = self.code_info(code)
        dev = _is_torch(source)
        F = source.shape[0]
        S = 1 if source.ndim == 2 else source.shape[2]
        assert source.shape[1] == k
        shape = (F, n) if source.ndim == 2 else (F, n, S)
        if dev:
            import torch
            out = torch.empty(shape, dtype=torch.uint8, device=source.device) if out is None else out
        else:
            source = np.ascontiguousarray(source, dtype=np.uint8)
            out = np.empty(shape, dtype=np.uint8) if out is None else out
        self._check(self._L.ldpc_amd_encode_batch(self._h, code, S, F, _ptr(source), _ptr(out), DEVICE_PTRS if dev else 0),
                    "encode_batch")
        return out

    # -- Reed-Solomon
    def rs_create(self, n, k):
        return self._check(self._L.ldpc_amd_rs_create(self._h, n, k), "rs_create")

    def rs_generator(self, rs, n, k):
        g = np.zeros((k, n), dtype=np.uint8)
        self._check(self._L.ldpc_amd_rs_generator(self._h, rs, g.ctypes.data), "rs_generator")
        return g

    def rs_encode(self, rs, n, k, source):
        dev = _is_torch(source)
        B = source.shape[0]
        S = 1 if source.ndim == 2 else source.shape[2]
        shape = (B, n) if source.ndim == 2 else (B, n, S)
        if dev:
            import torch
            out = torch.empty(shape, dtype=torch.uint8, device=source.device)
        else:
            source = np.ascontiguousarray(source, dtype=np.uint8)
            out = np.empty(shape, dtype=np.uint8)
        self._check(self._L.ldpc_amd_rs_encode_batch(self._h, rs, S, B, _ptr(source), _ptr(out), DEVICE_PTRS if dev else 0),
                    "rs_encode_batch")
        return out

    def rs_decode(self, rs, recv_idx, recv_val, out=None):
        """recv_idx [B,k] uint16 (0-based ascending), recv_val [B,k,S] or [B,k] -> msg like recv_val."""
        dev = _is_torch(recv_val)
        B = recv_val.shape[0]
        S = 1 if recv_val.ndim == 2 else recv_val.shape[2]
        if dev:
            import torch
            msg = torch.empty_like(recv_val) if out is None else out
        else:
            recv_idx = np.ascontiguousarray(recv_idx, dtype=np.uint16)
            recv_val = np.ascontiguousarray(recv_val, dtype=np.uint8)
            msg = np.empty_like(recv_val) if out is None else out
        self._check(self._L.ldpc_amd_rs_decode_batch(self._h, rs, S, B, _ptr(recv_idx), _ptr(recv_val), _ptr(msg),
                                                     DEVICE_PTRS if dev else 0), "rs_decode_batch")
        return msg

    def rs_bad_blocks(self):
        """Blocks of the last rs_decode whose positions were malformed (decoded to zeros).  Synchronises."""
        c = C.c_longlong(0)
        self._check(self._L.ldpc_amd_rs_bad_blocks(self._h, C.byref(c)), "rs_bad_blocks")
        return c.value

    # -- synthetic inputs on the device (torch tensors)
    def synth_source(self, seed, frame0, nframes, k, S, out):
        self._check(self._L.ldpc_amd_synth_source(self._h, seed, frame0, nframes, k, S, _ptr(out)), "synth_source")
        return out

    def synth_erasures_uniform(self, seed, frame0, nframes, n, per, out):
        self._check(self._L.ldpc_amd_synth_erasures_uniform(self._h, seed, frame0, nframes, n, float(per), _ptr(out)),
                    "synth_erasures_uniform")
        return out

    def synth_erasures_bursty(self, seed, frame0, nframes, n, alpha, beta, bias, out):
        self._check(self._L.ldpc_amd_synth_erasures_bursty(self._h, seed, frame0, nframes, n, float(alpha), float(beta),
                                                           float(bias), _ptr(out)), "synth_erasures_bursty")
        return out

    # -- FPGA harness trio (OpenCL/host/src/main.cpp:578-626)
    def data_in(self, nldpc, seed, per_numerator_div_64, code_ind, num_frames):
        self._check(self._L.ldpc_amd_data_in(self._h, None, nldpc, seed, per_numerator_div_64, code_ind, num_frames), "data_in")

    def ldpc_erasure_decoder(self, num_iter, code_ind):
        self._check(self._L.ldpc_amd_ldpc_erasure_decoder(self._h, num_iter, code_ind), "ldpc_erasure_decoder")

    def ldpc_erasure_decoder_perf_tests(self, num_iter, code_ind):
        self._check(self._L.ldpc_amd_ldpc_erasure_decoder_perf_tests(self._h, num_iter, code_ind), "ldpc_erasure_decoder_perf_tests")

    def fpga_frame_stats(self, num_frames):
        """(systematic erasures left, iterations) per frame of the last FPGA-style decoder call."""
        left = np.zeros(num_frames, dtype=np.int32)
        its = np.zeros(num_frames, dtype=np.int32)
        self._check(self._L.ldpc_amd_fpga_frame_stats(self._h, num_frames, left.ctypes.data, its.ctypes.data), "fpga_frame_stats")
        return left, its

    def data_out(self, code_ind, num_frames):
        st = ErrorType()
        self._check(self._L.ldpc_amd_data_out(self._h, None, code_ind, num_frames, C.byref(st)), "data_out")
        return st.num_LDPC_errors, st.num_RS_errors


# ---------------------------------------------------------------------------------------------------------
# Host-side wire format (include/ldpc_erasure_amd_wire.h): FEC header, packetiser, two-buffer reassembler.
# ---------------------------------------------------------------------------------------------------------
def fec_header_pack(fec_class, block, symbol):
    return int(load_library().ldpc_amd_fec_header_pack(fec_class, block, symbol))


def fec_header_unpack(word):
    a, b, c = C.c_uint(0), C.c_uint(0), C.c_uint(0)
    load_library().ldpc_amd_fec_header_unpack(C.c_uint64(word), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def fec_packetize(frames, fec_class=1, block0=0):
    """frames: uint8 [F][n][S] -> packets uint8 [F*n][8+S] in transmission order."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    F, n, S = frames.shape
    packets = np.zeros((F * n, 8 + S), dtype=np.uint8)
    if load_library().ldpc_amd_fec_packetize(frames.ctypes.data, F, n, S, fec_class, block0, packets.ctypes.data) != 0:
        raise LdpcAmdError("fec_packetize: bad arguments")
    return packets


class FecRx:
    """Two-buffer reassembler (OpenCL/device/ldpc_erasure_decoder_with_reordering_logic.cl:44-141,214-243).
    push(packet) / flush() return None or (block number, sym [n][S], erased [n])."""

    def __init__(self, n, k, S):
        self._L = load_library()
        h = C.c_void_p()
        if self._L.ldpc_amd_fec_rx_create(n, k, S, C.byref(h)) != 0:
            raise LdpcAmdError("fec_rx_create: bad arguments")
        self._h, self.n, self.k, self.S = h, n, k, S
        self._sym = np.zeros((n, S), dtype=np.uint8)
        self._er = np.zeros(n, dtype=np.uint8)

    def _ret(self, rc, blk):
        if rc < 0:
            raise LdpcAmdError("fec_rx: bad arguments")
        return (blk.value, self._sym.copy(), self._er.copy()) if rc == 1 else None

    def push(self, packet):
        packet = np.ascontiguousarray(packet, dtype=np.uint8)
        assert packet.size == 8 + self.S
        blk = C.c_int(-1)
        return self._ret(self._L.ldpc_amd_fec_rx_push(self._h, packet.ctypes.data, self._sym.ctypes.data, self._er.ctypes.data, C.byref(blk)), blk)

    def push_many(self, packets, max_blocks):
        """packets: uint8 [P][8+S].  Returns (blocks int32 [B], sym uint8 [B][n][S], erased uint8 [B][n], consumed)."""
        packets = np.ascontiguousarray(packets, dtype=np.uint8)
        assert packets.ndim == 2 and packets.shape[1] == 8 + self.S
        sym = np.zeros((max_blocks, self.n, self.S), dtype=np.uint8)
        er = np.zeros((max_blocks, self.n), dtype=np.uint8)
        blocks = np.zeros(max_blocks, dtype=np.int32)
        used = C.c_long(0)
        nb = self._L.ldpc_amd_fec_rx_push_many(self._h, packets.ctypes.data, packets.shape[0], sym.ctypes.data, er.ctypes.data,
                                               blocks.ctypes.data, max_blocks, C.byref(used))
        if nb < 0:
            raise LdpcAmdError("fec_rx_push_many: bad arguments")
        return blocks[:nb], sym[:nb], er[:nb], used.value

    def flush(self):
        blk = C.c_int(-1)
        return self._ret(self._L.ldpc_amd_fec_rx_flush(self._h, self._sym.ctypes.data, self._er.ctypes.data, C.byref(blk)), blk)

    @property
    def dropped(self):
        return int(self._L.ldpc_amd_fec_rx_dropped(self._h))

    def close(self):
        if self._h:
            self._L.ldpc_amd_fec_rx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_frames(nframes, nranks, rank):
    """(first, count) of rank's contiguous block: the library's C shard arithmetic (ldpc_amd_shard_frames)."""
    L = load_library()
    f0, cnt = C.c_int64(), C.c_int64()
    L.ldpc_amd_shard_frames(nframes, nranks, rank, C.byref(f0), C.byref(cnt))
    return f0.value, cnt.value


class Group:
    """The C-level multi-device layer (include/ldpc_erasure_amd_multi.h): nranks contexts, one host thread each; devices may
    repeat (several ranks on one device: how the layer is tested on a one-GPU box)."""

    def __init__(self, nranks, devices=None):
        self._L = load_library()
        h = C.c_void_p()
        dv = None if devices is None else (C.c_int * nranks)(*devices)
        rc = self._L.ldpc_amd_group_create(nranks, dv, C.byref(h))
        if rc != OK:
            raise LdpcAmdError(f"ldpc_amd_group_create({nranks}) = {rc}: {self._L.ldpc_amd_last_error(None).decode()}")
        self._h = h
        self.nranks = nranks

    def close(self):
        if getattr(self, "_h", None):
            self._L.ldpc_amd_group_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what):
        if rc < 0:
            raise LdpcAmdError(f"{what} = {rc}: {self._L.ldpc_amd_group_last_error(self._h).decode()}")
        return rc

    def device(self, rank):
        return self._L.ldpc_amd_group_device(self._h, rank)

    def load_builtin_code(self, code_ind, coef_seed):
        return self._check(self._L.ldpc_amd_group_load_builtin_code(self._h, code_ind, coef_seed), "group_load_builtin_code")

    def register_code(self, code):
        rp = np.ascontiguousarray(code.row_ptr, dtype=np.uint32)
        cols = np.ascontiguousarray(code.cols, dtype=np.uint16)
        coefs = np.ascontiguousarray(code.coefs, dtype=np.uint8)
        return self._check(self._L.ldpc_amd_group_register_code(self._h, code.n, code.k, rp.ctypes.data, cols.ctypes.data, coefs.ctypes.data),
                           "group_register_code")

    def decode(self, code, sym, erased, max_sweeps=10, do_ml=1):
        """Host arrays sym [F,n,S] or [F,n], erased [F,n] -> (out, sweeps, residual, status), frames sharded over the ranks."""
        sym = np.ascontiguousarray(sym, dtype=np.uint8)
        erased = np.ascontiguousarray(erased, dtype=np.uint8)
        F = sym.shape[0]
        S = 1 if sym.ndim == 2 else sym.shape[2]
        out = np.empty_like(sym)
        sw, res, st = (np.empty(F, dtype=np.int32) for _ in range(3))
        self._check(self._L.ldpc_amd_group_decode_batch(self._h, code, S, F, sym.ctypes.data, erased.ctypes.data, max_sweeps, do_ml, out.ctypes.data,
                                                        sw.ctypes.data, res.ctypes.data, st.ctypes.data), "group_decode_batch")
        return out, sw, res, st

    def decode_resident(self, code, S, nframes, sym, erased, out, words, gathered_words=None, gathered_out=None, max_sweeps=10, do_ml=1):
        """Per-rank lists of torch CUDA tensors (rank r's shard on its device); gathered_*: tensors on rank 0's device or None.
        Returns (decode_ms, gather_ms)."""
        arr = lambda ts: (C.c_void_p * self.nranks)(*[t.data_ptr() if t is not None else None for t in ts])   # noqa: E731
        dms, gms = C.c_double(0), C.c_double(0)
        self._check(self._L.ldpc_amd_group_decode_resident(self._h, code, S, nframes, arr(sym), arr(erased), max_sweeps, do_ml, arr(out), arr(words),
                                                           None if gathered_words is None else gathered_words.data_ptr(),
                                                           None if gathered_out is None else gathered_out.data_ptr(), C.byref(dms), C.byref(gms)),
                    "group_decode_resident")
        return dms.value, gms.value

    def fpga_run(self, nldpc, seed, per64, code_ind, num_frames, num_iter, perf_tests_body=False):
        st = ErrorType()
        self._check(self._L.ldpc_amd_group_fpga_run(self._h, nldpc, seed, per64, code_ind, num_frames, num_iter, int(perf_tests_body), C.byref(st)),
                    "group_fpga_run")
        return st.num_LDPC_errors, st.num_RS_errors

    def bench_resident(self, code_ind, coef_seed, S, frames_per_rank, per=0.10, max_sweeps=10, steps=5):
        r = (C.c_double * 4)()
        self._check(self._L.ldpc_amd_group_bench_resident(self._h, code_ind, coef_seed, S, frames_per_rank, per, max_sweeps, steps, r), "group_bench_resident")
        return dict(zip(("frames_per_s", "decode_ms_per_step", "gather_ms", "verified"), list(r)))
