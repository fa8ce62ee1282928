"""ctypes binding of libseekr_hip.so (see include/seekr_hip.h) plus thin RAII wrappers.

There is deliberately NO fallback: if the HIP library is missing or no MI355X is visible
every compute entry point raises.  `import seekr_amd` itself stays importable on a CPU-only
box so that the build step and the host-logic tests can run there.
"""
import atexit
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libseekr_hip.so")
DIAG_LIB_PATH = os.path.join(_HERE, "libseekr_hip_diag.so")

SKR_OK = 0
F32, F64, U32 = 0, 1, 2
LOG2_NONE, LOG2_PRE, LOG2_POST = 0, 1, 2
PREC_FP32, PREC_BF16X3, PREC_F64, PREC_BF16X4, PREC_F16X3, PREC_F16F8 = 0, 1, 2, 3, 4, 5
LOG2_CODES = {"Log2.none": LOG2_NONE, "Log2.pre": LOG2_PRE, "Log2.post": LOG2_POST}
PRECISIONS = {"fp32": PREC_FP32, "bf16x3": PREC_BF16X3, "f16x3": PREC_F16X3, "f64": PREC_F64,  # (bf16x4 = 3: retired in round 5)
              "f16f8": PREC_F16F8}  # f16f8: opt-in, two product-units per k; degrades to f16x3 by itself
_NP_OF = {F32: np.float32, F64: np.float64, U32: np.uint32}
_CODE_OF = {np.dtype(np.float32): F32, np.dtype(np.float64): F64, np.dtype(np.uint32): U32}

_p = C.c_void_p
_i64 = C.c_int64
_int = C.c_int

# name -> (restype, argtypes); every symbol include/seekr_hip.h declares
SIGNATURES = {
    "skr_last_error": (C.c_char_p, []),
    "skr_abi_version": (_int, []),
    "skr_device_count": (_int, [C.POINTER(_int)]),
    "skr_ctx_create": (_int, [_int, C.POINTER(_p)]),
    "skr_ctx_destroy": (_int, [_p]),
    "skr_ctx_sync": (_int, [_p]),
    "skr_ctx_mem_info": (_int, [_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "skr_ctx_device": (_int, [_p, C.POINTER(_int)]),
    "skr_ctx_reload_knobs": (_int, [_p]),
    "skr_prof_enable": (_int, [_p, _int]),
    "skr_prof_reset": (_int, [_p]),
    "skr_prof_query": (_int, [_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "skr_prof_names": (_int, [_p, C.c_char_p, _i64]),
    "skr_mat_create": (_int, [_p, _i64, _i64, _int, C.POINTER(_p)]),
    "skr_mat_free": (_int, [_p]),
    "skr_mat_shape": (_int, [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_int)]),
    "skr_mat_upload": (_int, [_p, _p, _i64, _i64]),
    "skr_mat_download": (_int, [_p, _p, _i64, _i64]),
    "skr_mat_fill_zero": (_int, [_p]),
    "skr_host_register": (_int, [_int, _p, C.c_size_t]),
    "skr_host_unregister": (_int, [_p]),
    "skr_ctx_mark": (_int, [_p, C.POINTER(_i64)]),
    "skr_ctx_mark_release": (_int, [_p, _i64]),
    "skr_mat_download_at": (_int, [_p, _p, _i64, _i64, _i64]),
    "skr_npy_create": (_int, [C.c_char_p, _int, _i64, _i64, C.POINTER(_i64)]),
    "skr_mat_write_rows_at": (_int, [_p, _i64, _i64, C.c_char_p, _i64, _i64]),
    "skr_fasta_open": (_int, [C.c_char_p, C.POINTER(_p)]),
    "skr_fasta_free": (_int, [_p]),
    "skr_fasta_info": (_int, [_p, C.POINTER(_i64), C.POINTER(_i64)]),
    "skr_fasta_lengths": (_int, [_p, _p]),
    "skr_fasta_headers": (_int, [_p, C.c_char_p, _i64, C.POINTER(_i64)]),
    "skr_fasta_pack": (_int, [_p, _p, _i64, _i64, C.c_char_p, C.POINTER(_p)]),
    "skr_operand_x8_pair_bound": (_int, [_p, _p, C.POINTER(C.c_double), C.POINTER(_int)]),
    "skr_mat_view": (_int, [_p, _i64, _i64, C.POINTER(_p)]),
    "skr_mat_device_ptr": (_int, [_p, C.POINTER(_p)]),
    "skr_seqs_pack": (_int, [_p, _p, _p, _i64, C.c_char_p, C.POINTER(_p)]),
    "skr_seqs_from_fasta": (_int, [_p, C.c_char_p, C.c_char_p, C.POINTER(_p)]),
    "skr_seqs_free": (_int, [_p]),
    "skr_seqs_info": (_int, [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "skr_seqs_lengths": (_int, [_p, _p]),
    "skr_seqs_headers": (_int, [_p, C.c_char_p, _i64, C.POINTER(_i64)]),
    "skr_count_u32": (_int, [_p, _p, _int, _p]),
    "skr_count_per_kb": (_int, [_p, _p, _int, _int, _p]),
    "skr_colsum_seq": (_int, [_p, _p, _p, _p, _int, _p]),
    "skr_colsum_seq_colmin": (_int, [_p, _p, _p, _p]),
    "skr_chain_create": (_int, [_p, _i64, C.POINTER(_p)]),
    "skr_chain_export": (_int, [_p, C.c_char_p]),
    "skr_chain_connect": (_int, [_p, _int, _int, C.c_char_p]),
    "skr_chain_connect_local": (_int, [_p, _int, _int, C.POINTER(_p)]),
    "skr_colsum_seq_chain": (_int, [_p, _p, _p, _p, _int, _p, _p, _int]),
    "skr_chain_result": (_int, [_p, _p]),
    "skr_chain_check": (_int, [_p, C.POINTER(_int)]),
    "skr_chain_free": (_int, [_p]),
    "skr_vec_finish": (_int, [_p, _p, _i64, _int]),
    "skr_min_nan": (_int, [_p, _p, _p, _p, C.POINTER(C.c_float), C.POINTER(_int)]),
    "skr_apply": (_int, [_p, _p, _int, _p, _p, _int, C.c_float, _p, C.POINTER(_int)]),
    "skr_normalize": (_int, [_p, _p, _int, _int, _p, _int, _p, _p, _p, C.POINTER(_int)]),
    "skr_row_standardize": (_int, [_p, _p, _p]),
    "skr_pearson_gemm": (_int, [_p, _p, _p, _int, _int, _p, _i64, _i64]),
    "skr_pearson": (_int, [_p, _p, _p, _int, _int, _p]),
    "skr_count_generic": (_int, [_p, _p, _p, _i64, C.c_char_p, _int, _int, _int, _p]),
    "skr_aseqs_create": (_int, [_p, _p, _p, _i64, C.POINTER(_p)]),
    "skr_aseqs_free": (_int, [_p]),
    "skr_count_generic_dev": (_int, [_p, _p, C.c_char_p, _int, _int, _int, _p]),
    "skr_host_get_counts": (_int, [_p, _p, _int, _int, _int, _p, _int, _int, _p, _int, _p, _p, _p, C.POINTER(_int)]),
    "skr_host_pearson": (_int, [_p, _p, _i64, _p, _i64, _i64, _int, _int, _int, _p]),
    "skr_operand_create": (_int, [_p, _i64, _i64, _int, C.POINTER(_p)]),
    "skr_operand_free": (_int, [_p]),
    "skr_operand_view": (_int, [_p, _i64, _i64, C.POINTER(_p)]),
    "skr_operand_as_mat": (_int, [_p, C.POINTER(_p)]),
    "skr_operand_fill": (_int, [_p, _p, _p, _p, _int, C.c_float, _p, _int, _p, C.POINTER(_int)]),
    "skr_operand_kind": (_int, [_p, C.POINTER(_int)]),
    "skr_operand_adopt_layout": (_int, [_p, _p]),
    "skr_operand_coherent": (_int, [_p, _int, C.POINTER(_int)]),
    "skr_operand_x8_stats": (_int, [_p, _int, C.POINTER(C.c_float)]),
    "skr_pearson_gemm_op": (_int, [_p, _p, _p, _int, _p, _i64, _i64]),
    "skr_pearson_gemm_op_rows": (_int, [_p, _p, _p, _i64, _p, _i64]),
    "skr_pearson_gemm_f64": (_int, [_p, _p, _p, _i64, _int, _p, _i64, _i64]),
    "skr_pearson_gemm_op_mirror": (_int, [_p, _p, _p, _p, _i64, _i64, _p, _i64, _i64]),
    "skr_threshold_zero_diag": (_int, [_p, _p, C.c_float, _i64]),
    "skr_triu_flatten": (_int, [_p, _p, _i64, _p]),
    "skr_gather_f32": (_int, [_p, _p, _p, _i64, _p]),
    "skr_empirical_pvalues": (_int, [_p, _p, _p, _i64, _p]),
    "skr_parametric_pvalues": (_int, [_p, _p, C.c_char_p, C.POINTER(C.c_double), _int, _p]),
    "skr_edges": (_int, [_p, _p, _i64, _i64, _i64, _i64, _i64, C.c_float, _int, _p, _p, _p, C.POINTER(_i64)]),
    "skr_pearson_gemm_edges": (_int, [_p, _p, _p, _p, _i64, _i64, C.c_float, _int, _p, _p, _p, C.POINTER(_i64)]),
    "skr_pearson_gemm_edges_needs_scratch": (_int, [_p, _p, _p, C.POINTER(_int)]),
    "skr_topk_rows": (_int, [_p, _p, _i64, _i64, _i64, _i64, _i64, _int, _p, _p]),
    "skr_mat_save_npy": (_int, [_p, _p, _int, C.c_char_p]),
    "skr_mat_save_csv": (_int, [_p, _p, _int, _int, C.c_char_p]),
    "skr_host_save_npy": (_int, [_p, _int, _i64, _i64, _int, C.c_char_p]),
    "skr_host_save_csv": (_int, [_p, _int, _i64, _i64, _int, _int, C.c_char_p]),
    "skr_mat_save_csv_labelled": (_int, [_p, _p, C.c_char_p, C.c_char_p, _int, C.c_char_p]),
    "skr_host_save_csv_labelled": (_int, [_p, _int, _i64, _i64, C.c_char_p, C.c_char_p, _int, C.c_char_p]),
    "skr_csv_read": (_int, [C.c_char_p, _int, C.POINTER(_p)]),
    "skr_csv_shape": (_int, [_p, C.POINTER(_i64), C.POINTER(_i64)]),
    "skr_csv_values": (_int, [_p, _p]),
    "skr_csv_labels": (_int, [_p, _int, C.c_char_p, _i64, C.POINTER(_i64)]),
    "skr_csv_free": (_int, [_p]),
    "skr_event_record": (_int, [_p, _int, C.POINTER(_p)]),
    "skr_event_wait": (_int, [_p, _int, _p]),
    "skr_event_free": (_int, [_p]),
    "skr_peer_copy_rows": (_int, [_p, _i64, _p, _i64, _i64]),
    "skr_comm_unique_id": (_int, [C.c_char_p]),
    "skr_comm_init": (_int, [_p, _int, _int, C.c_char_p]),
    "skr_comm_destroy": (_int, [_p]),
    "skr_comm_barrier": (_int, [_p]),
    "skr_comm_sendrecv": (_int, [_p, _p, _i64, _i64, _int, _p, _i64, _i64, _int, C.POINTER(_i64)]),
    "skr_comm_allgather_rows": (_int, [_p, _p, _p, C.POINTER(_i64), C.POINTER(_i64)]),
    "skr_comm_exchange": (_int, [_p, _int, _p, _p, _p, _p, _p, _p, _p, _p, C.POINTER(_i64)]),
    "skr_comm_wait": (_int, [_p, _i64]),
    "skr_comm_allreduce_f64": (_int, [_p, C.POINTER(C.c_double), _int, _int]),
    "skr_host_colstat": (_int, [_p, _p, _i64, _i64, _int, _int, _p]),
    "skr_host_colstat_colmajor": (_int, [_p, _p, _i64, _i64, _int, _int, _p]),
    "skr_host_apply": (_int, [_p, _p, _i64, _i64, _int, _int, _p, _int, _p, _int, C.POINTER(_int)]),
}
# libseekr_hip_diag.so only (tools/gemm_diag.py sets LIB_PATH to it before the first call)
DIAG_SIGNATURES = {
    "skr_gemm_diag_mode": (_int, [_p, _int]),
    "skr_gemm_diag_read": (_int, [_p, _p, _i64, C.POINTER(_i64)]),
}

_lib = None
_lock = threading.Lock()
_shutdown = False


def _mark_shutdown():
    # at interpreter exit the HIP runtime reclaims everything; explicit frees in arbitrary
    # finaliser order would race its own teardown
    global _shutdown
    _shutdown = True


atexit.register(_mark_shutdown)


class SeekrHipError(RuntimeError):
    """A HIP / RCCL runtime failure inside libseekr_hip."""


class FastaNeedsText(Exception):
    """The native FASTA parser declined the file (SKR_ERR_FASTA_TEXT: a byte >= 0x80).  The reference opens the file in
    text mode (fasta_reader.py:44), so characters — not bytes — are stripped, upper-cased and counted in len(seq); the
    caller reads such a file with `seekr_amd.fasta_reader.Reader` and packs the strings."""


IPC_VAR = "HSA_ENABLE_IPC_MODE_LEGACY"
_ipc_env_at_load = "(library not loaded)"  # what IPC_VAR held when the HIP runtime came into the process


def names_several_devices(spec=None):
    """Does SEEKR_DEVICES (or `spec`) ask for more than one GPU?  Answered from the text alone — no HIP call."""
    spec = (os.environ.get("SEEKR_DEVICES", "") if spec is None else spec).strip().lower()
    return spec == "all" or len([t for t in spec.split(",") if t.strip()]) > 1


def prepare_runtime_env():
    """What must stand in the environment BEFORE the first HIP call of the process.  RCCL's peer-to-peer set-up between the
    GPU threads of SEEKR_DEVICES needs dmabuf IPC (HSA_ENABLE_IPC_MODE_LEGACY=0: the driver of these nodes supports no
    other), and the HIP / HSA runtime reads its environment once, when it starts — so the variable is set here, at
    `import seekr_amd` and again right before libseekr_hip.so is loaded, whenever SEEKR_DEVICES names several devices.
    A value the user exported is left alone.  Nothing to do once the library is in the process."""
    if _lib is None and names_several_devices():
        os.environ.setdefault(IPC_VAR, "0")


def ipc_env_at_load():
    """The value of HSA_ENABLE_IPC_MODE_LEGACY the HIP runtime started with (None = unset)."""
    return _ipc_env_at_load


def lib():
    """Load libseekr_hip.so once; raise ImportError (never fall back) when it is absent."""
    global _lib, _ipc_env_at_load
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "seekr_amd: {} is missing — build it with `python -m seekr_amd.build` "
                "(hipcc, gfx950). There is no CPU fallback.".format(LIB_PATH))
        prepare_runtime_env()
        _ipc_env_at_load = os.environ.get(IPC_VAR)
        handle = C.CDLL(LIB_PATH)
        sigs = dict(SIGNATURES, **DIAG_SIGNATURES) if LIB_PATH == DIAG_LIB_PATH else SIGNATURES
        for name, (res, args) in sigs.items():
            fn = getattr(handle, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        if handle.skr_abi_version() != 1:
            raise ImportError("seekr_amd: libseekr_hip.so has ABI {}, expected 1".format(handle.skr_abi_version()))
        _lib = handle
    return _lib


_EXC = {
    -1: ValueError,
    -2: SeekrHipError,
    -3: MemoryError,
    -4: NotImplementedError,
    -5: ZeroDivisionError,
    -6: SeekrHipError,
    -7: OSError,
    -8: IndexError,
    -9: AssertionError,
    -10: FastaNeedsText,
}


def check(rc):
    if rc == SKR_OK:
        return
    msg = lib().skr_last_error().decode("utf-8", "replace")
    raise _EXC.get(rc, SeekrHipError)(msg)


def device_count():
    n = _int(0)
    rc = lib().skr_device_count(C.byref(n))
    return n.value if rc == SKR_OK else 0


_last_device = None  # the device of the newest Context: where the result pool registers its pages (no Context, no GPU: no registering)


class Context:
    """One GPU + one stream (skr_ctx)."""

    def __init__(self, device=0):
        global _last_device
        self._h = _p()
        self.device = device
        check(lib().skr_ctx_create(int(device), C.byref(self._h)))
        _last_device = int(device)

    def close(self):
        if getattr(self, "_h", None) and not _shutdown:
            lib().skr_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def sync(self):
        check(lib().skr_ctx_sync(self._h))

    def mark(self):
        """The point the compute stream has reached (skr_ctx_mark): Matrix.to_numpy_at / write_rows_at wait for it only."""
        m = _i64(-1)
        check(lib().skr_ctx_mark(self._h, C.byref(m)))
        return m.value

    def mark_release(self, mark):
        """Hand back a mark that will not be passed to a download after all (skr_ctx_mark_release; always safe)."""
        if mark is not None and getattr(self, "_h", None) and not _shutdown:
            lib().skr_ctx_mark_release(self._h, int(mark))

    def mem_info(self):
        """(free, total) bytes of device memory."""
        f, t = C.c_uint64(0), C.c_uint64(0)
        check(lib().skr_ctx_mem_info(self._h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def reload_knobs(self):
        """Re-read the SEEKR_GEMM_* / SEEKR_COUNT_* A/B switches from os.environ (they are otherwise read once, at creation)."""
        check(lib().skr_ctx_reload_knobs(self._h))

    # ---- profiling -----------------------------------------------------------------
    def prof_enable(self, on=True):
        check(lib().skr_prof_enable(self._h, 1 if on else 0))

    def prof_reset(self):
        check(lib().skr_prof_reset(self._h))

    def prof_query(self, name):
        """(total ms, launches) recorded under exactly `name` (prof_names lists them)."""
        ms, cnt = C.c_double(0), _i64(0)
        check(lib().skr_prof_query(self._h, name.encode(), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def prof_names(self):
        buf = C.create_string_buffer(4096)
        check(lib().skr_prof_names(self._h, buf, 4096))
        return [s for s in buf.value.decode().split("\n") if s]

    # ---- factories ------------------------------------------------------------------
    def empty(self, rows, cols, dtype=np.float32):
        return Matrix(self, rows, cols, dtype)

    def zeros(self, rows, cols, dtype=np.float32):
        m = Matrix(self, rows, cols, dtype)
        check(lib().skr_mat_fill_zero(m._h))
        return m

    def from_numpy(self, a):
        a = np.ascontiguousarray(a)
        if a.ndim == 1:
            a = a.reshape(1, -1)
        if a.ndim != 2:
            raise ValueError("expected a 1-D or 2-D array")
        m = Matrix(self, a.shape[0], a.shape[1], a.dtype)
        m.upload(a)
        return m

    def pack(self, seqs, alphabet="AGTC"):
        return PackedSeqs.from_strings(self, seqs, alphabet)

    def pack_fasta(self, path, alphabet="AGTC"):
        return PackedSeqs.from_fasta(self, path, alphabet)


_default_ctx = {}
# A skr_ctx is ONE stream plus its scratch (flag words, workspaces): calls through one ctx from several host threads are
# not safe against each other inside the library.  The drop-in API (BasicCounter's computing methods, pearson(),
# pearson_to_file()) works on the process-wide default context, so its entry points take this lock: two Python threads
# may call them freely, the calls run one after the other (the reference is "not re-entrant, no globals": SURVEY 8b).
# Code that drives device handles itself (Context / Matrix / Operand, seekr_amd.consumers) from several threads holds
# it too, or gives every thread its own Context.
API_LOCK = threading.RLock()


def api_call(fn):
    """Decorator: run the function under API_LOCK (see above)."""
    import functools

    @functools.wraps(fn)
    def locked(*args, **kwargs):
        with API_LOCK:
            return fn(*args, **kwargs)
    return locked



def default_device():
    """$SEEKR_DEVICE, else the one device SEEKR_DEVICES names, else LOCAL_RANK, else 0."""
    dev = os.environ.get("SEEKR_DEVICE")
    if dev is None:
        listed = [t for t in os.environ.get("SEEKR_DEVICES", "").split(",") if t.strip()]
        dev = listed[0] if len(listed) == 1 and listed[0].strip().isdigit() else os.environ.get("LOCAL_RANK", "0")
    return int(dev)


def default_context():
    """Process-wide context on default_device()."""
    dev = default_device()
    ctx = _default_ctx.get(dev)
    if ctx is None:
        ctx = _default_ctx[dev] = Context(dev)
    return ctx


# ----------------------------------------------------------------------------- result memory --
class _Slab:
    """One block of host memory the pool owns: a uint8 array and whether its pages are registered with the HIP runtime."""
    __slots__ = ("arr", "registered", "_map")

    def __init__(self, nbytes):
        # an anonymous mapping of its own — whole pages that belong to this slab alone (a malloc'ed block of a few MiB may
        # sit on the heap and share its first and last page with other objects: not something to page-lock and unlock);
        # fresh pages: the first copy into them costs what it always did
        import mmap
        self._map = mmap.mmap(-1, max(int(nbytes), 1))
        self.arr = np.frombuffer(self._map, dtype=np.uint8, count=int(nbytes))
        self.registered = False

    @property
    def nbytes(self):
        return self.arr.nbytes

    def register(self):
        """Page-lock the (by now touched) pages: 5-7 ms for 576 MB, once; every later copy into them is plain DMA at the link's
        rate — without it the rate depends on the state of the runtime's own pinning cache (10 or 19 ms for the same 549 MB
        download, measured in one process).  Quietly skipped where it cannot be done (no GPU, locked-memory limit)."""
        if (not self.registered and not _shutdown and _last_device is not None
                and os.environ.get("SEEKR_RESULT_POOL_REGISTER", "1") != "0"):
            try:  # on the device of the newest Context: this may be a finaliser on a thread that has never chosen a device
                self.registered = lib().skr_host_register(int(_last_device), self.arr.ctypes.data_as(_p), self.arr.nbytes) == SKR_OK
            except Exception:  # noqa: BLE001
                self.registered = False

    def release(self):
        if self.registered and not _shutdown:
            try:
                lib().skr_host_unregister(self.arr.ctypes.data_as(_p))
            except Exception:  # noqa: BLE001
                pass
        self.registered = False

    def __del__(self):
        self.release()  # never hand registered pages back to the allocator


class _Lease:
    """What a pooled result array rests on (its `.base`): when the array and every view of it are gone, the memory goes
    back to the pool instead of to the operating system."""
    __slots__ = ("_pool", "_slab", "__array_interface__")

    def __init__(self, pool, slab, shape, dtype):
        self._pool, self._slab = pool, slab
        self.__array_interface__ = {"data": (slab.arr.ctypes.data, False), "shape": tuple(int(d) for d in shape),
                                    "typestr": np.dtype(dtype).str, "version": 3}

    def __del__(self):
        try:
            self._pool._give_back(self._slab)
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class HostPool:
    """Host memory for the arrays the API returns (pearson()'s r, get_counts()'s matrix), kept between calls.

    A device-to-host copy into pages the process has never touched runs at 22-24 GB/s on the GPU box — the kernel
    zero-fills every destination page inside the runtime's copy thread — and at up to 55 GB/s into pages it has written
    before.  numpy hands a large array's pages back to the operating system when the array dies, so
    every call pays for fresh ones.  Here the FIRST result of a size is a plain allocation (no slower than before: nothing
    is locked or touched up front); when the caller drops it, its pages stay with the process and are registered with the
    HIP runtime (hipHostRegister: milliseconds for touched pages), and the next result of about that size lands in them by
    plain DMA at the link's rate: pearson() host to host at 12 000 rows 57 -> 16 ms, FASTA -> host counts 78 -> 38 ms.

    SEEKR_RESULT_POOL_MB caps what is kept (default: an eighth of the machine's memory, at most 16 GiB; 0 switches the
    pool off); a result larger than the cap, or smaller than 1 MiB, is an ordinary array.  What the caller gets is an
    ordinary ndarray whose `.base` is the lease (`flags.owndata` is False, as for any view)."""
    MIN_BYTES = 1 << 20

    def __init__(self):
        self._lock = threading.RLock()  # re-entrant: a lease may die (collector) while this thread is inside empty()
        self._free = []                 # _Slab objects, touched, not in use
        self._kept = 0
        self._cap = None
        self.stats = {"fresh": 0, "reused": 0, "kept_bytes": 0, "dropped": 0, "registered": 0}

    def cap_bytes(self):
        if self._cap is None:
            env = os.environ.get("SEEKR_RESULT_POOL_MB", "").strip()
            if env:
                self._cap = max(0, int(float(env))) << 20
            else:
                try:
                    total = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
                except (ValueError, OSError):
                    total = 8 << 30
                self._cap = min(16 << 30, total // 8)
        return self._cap

    def empty(self, shape, dtype):
        shape = tuple(int(d) for d in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        dtype = np.dtype(dtype)
        n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if n < self.MIN_BYTES or n > self.cap_bytes():
            return np.empty(shape, dtype)
        slab = None
        with self._lock:
            best = None
            for i, cand in enumerate(list(self._free)):  # the smallest kept slab that holds n without wasting half of itself
                if n <= cand.nbytes <= 2 * n and (best is None or cand.nbytes < self._free[best].nbytes):
                    best = i
            if best is not None:
                slab = self._free.pop(best)
                self._kept -= slab.nbytes
                self.stats["reused"] += 1
                self.stats["kept_bytes"] = self._kept
        if slab is None:
            slab = _Slab(n)
            self.stats["fresh"] += 1
        return np.asarray(_Lease(self, slab, shape, dtype))

    def _give_back(self, slab):
        if _shutdown:
            return
        with self._lock:
            if self._kept + slab.nbytes <= self.cap_bytes():
                if not slab.registered:
                    slab.register()
                    self.stats["registered"] += int(slab.registered)
                self._free.append(slab)
                self._kept += slab.nbytes
            else:
                slab.release()
                self.stats["dropped"] += 1
            self.stats["kept_bytes"] = self._kept

    def clear(self):
        with self._lock:
            for slab in self._free:
                slab.release()
            self._free, self._kept = [], 0
            self.stats["kept_bytes"] = 0


host_pool = HostPool()


class Matrix:
    """Row-major device matrix (skr_mat)."""

    def __init__(self, ctx, rows, cols, dtype=np.float32, _view_of=None, _row0=0):
        code = _CODE_OF.get(np.dtype(dtype))
        if code is None:
            raise TypeError("device matrices are float32, float64 or uint32 (got {})".format(dtype))
        self.ctx = ctx
        self.rows, self.cols, self.dtype = int(rows), int(cols), np.dtype(dtype)
        self._h = _p()
        self._parent = _view_of  # keeps the owner alive
        if _view_of is None:
            check(lib().skr_mat_create(ctx._h, self.rows, self.cols, code, C.byref(self._h)))
        else:
            check(lib().skr_mat_view(_view_of._h, int(_row0), self.rows, C.byref(self._h)))

    def view(self, row0, nrows):
        """Non-owning view of rows [row0, row0+nrows)."""
        return Matrix(self.ctx, nrows, self.cols, self.dtype, _view_of=self, _row0=row0)

    @property
    def shape(self):
        return (self.rows, self.cols)

    def free(self):
        # a closed Context has been destroyed on the C side: its handles must not be touched
        if getattr(self, "_h", None) and not _shutdown and getattr(self.ctx, "_h", None):
            lib().skr_mat_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass

    def upload(self, a, row0=0):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.ndim == 1:
            a = a.reshape(1, -1)
        if a.shape[1] != self.cols:
            raise ValueError("column count mismatch")
        check(lib().skr_mat_upload(self._h, a.ctypes.data_as(_p), int(row0), a.shape[0]))

    def to_numpy_at(self, mark, out, row0=0):
        """Rows [row0, row0 + len(out)) into `out` on the copy stream, as soon as the compute stream has passed `mark`
        (Context.mark(); work enqueued after the mark is not waited for): skr_mat_download_at."""
        assert out.flags.c_contiguous and out.dtype == self.dtype and out.ndim == 2 and out.shape[1] == self.cols
        check(lib().skr_mat_download_at(self._h, out.ctypes.data_as(_p), int(row0), int(out.shape[0]), int(mark)))
        return out

    def write_rows_at(self, mark, path, file_offset, row0, nrows):
        """Rows [row0, row0 + nrows) to byte `file_offset` of the file (skr_mat_write_rows_at), behind `mark`."""
        check(lib().skr_mat_write_rows_at(self._h, int(row0), int(nrows), os.fsencode(path), int(file_offset), int(mark)))

    def to_numpy(self, row0=0, nrows=None, out=None):
        nrows = self.rows - row0 if nrows is None else nrows
        if out is None:
            out = host_pool.empty((nrows, self.cols), self.dtype)
        assert out.flags.c_contiguous and out.dtype == self.dtype and out.shape == (nrows, self.cols)
        check(lib().skr_mat_download(self._h, out.ctypes.data_as(_p), int(row0), int(nrows)))
        return out

    def vector(self):
        return self.to_numpy().reshape(-1)

    def device_ptr(self):
        ptr = _p()
        check(lib().skr_mat_device_ptr(self._h, C.byref(ptr)))
        return ptr.value


class Operand:
    """Rows prepared for the Pearson contraction (skr_operand)."""

    def __init__(self, ctx, rows, cols, precision=PREC_F16X3, _view_of=None, _row0=0):
        self.ctx, self.rows, self.cols, self.precision = ctx, int(rows), int(cols), int(precision)
        self._h = _p()
        self._parent = _view_of
        self._mat = None
        if _view_of is None:
            check(lib().skr_operand_create(ctx._h, self.rows, self.cols, self.precision, C.byref(self._h)))
        else:
            check(lib().skr_operand_view(_view_of._h, int(_row0), self.rows, C.byref(self._h)))

    def view(self, row0, nrows):
        return Operand(self.ctx, nrows, self.cols, self.precision, _view_of=self, _row0=row0)

    @property
    def kind(self):
        """0 = float32 layout (fp32 kernel), 1 = bf16 halves, 2 = fp16 halves, 3 = fp16 hi lines + fp8 cross lines (f16f8)
        (see skr_operand_kind)."""
        k = _int(0)
        check(lib().skr_operand_kind(self._h, C.byref(k)))
        return k.value

    @property
    def coherent(self):
        """True when the rows are mostly one repeated value (the contraction then restarts its accumulators more often)."""
        v = _int(0)
        check(lib().skr_operand_coherent(self._h, 0, C.byref(v)))
        return bool(v.value)

    @coherent.setter
    def coherent(self, flag):
        v = _int(1 if flag else 0)
        check(lib().skr_operand_coherent(self._h, 1, C.byref(v)))

    @property
    def x8_stats(self):
        """f16f8 layout only (zeros otherwise): the largest |row mean| of hi - 128 h8, of lo and of lo - l8 / 16 over the
        rows of this operand (skr_operand_x8_stats) — what the routing rule on the row means is evaluated from."""
        v = (C.c_float * 3)()
        check(lib().skr_operand_x8_stats(self._h, 0, v))
        return tuple(float(t) for t in v)

    @x8_stats.setter
    def x8_stats(self, values):
        v = (C.c_float * 3)(*[float(t) for t in values])
        check(lib().skr_operand_x8_stats(self._h, 1, v))

    def x8_pair_bound(self, other=None):
        """(bound, ok): the error of a cell of r that the row means of the rounding residues allow when rows of this
        operand meet rows of `other` (None: its own rows) in the f16f8 layout, from the maxima the operands carry NOW
        (x8_stats; a multi-GPU caller sets the all-reduced ones first), and whether the fill's limit admits it — the
        library's own rule (skr_operand_x8_pair_bound), never re-derived here.  (0, True) for any other layout."""
        bound, ok = C.c_double(0), _int(1)
        check(lib().skr_operand_x8_pair_bound(self._h, (other or self)._h, C.byref(bound), C.byref(ok)))
        return bound.value, bool(ok.value)

    def adopt_layout(self, like):
        """Tag this buffer (a receive buffer) with the storage kind of `like`."""
        check(lib().skr_operand_adopt_layout(self._h, like._h))
        return self

    def as_matrix(self):
        """float32-typed view of the storage, for the RCCL send/recv entry points."""
        if self._mat is None:
            h = _p()
            check(lib().skr_operand_as_mat(self._h, C.byref(h)))
            m = Matrix.__new__(Matrix)
            m.ctx, m.rows, m.cols, m.dtype = self.ctx, self.rows, 32 * ((self.cols + 31) // 32), np.dtype(np.float32)
            m._h, m._parent = h, self
            self._mat = m
        return self._mat

    def free(self):
        if getattr(self, "_h", None) and not _shutdown and getattr(self.ctx, "_h", None):
            if self._mat is not None:
                self._mat.free()
            lib().skr_operand_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


def _h(m):
    return m._h if m is not None else None


class FastaFile:
    """A FASTA file parsed into host memory with the reference reader's semantics (skr_fasta): read once, then packed
    onto one GPU or range by range onto several (pack() only reads: the GPUs of a node pack their ranges at once)."""

    def __init__(self, path):
        self._h = _p()
        check(lib().skr_fasta_open(os.fsencode(path), C.byref(self._h)))
        n, tot = _i64(0), _i64(0)
        check(lib().skr_fasta_info(self._h, C.byref(n), C.byref(tot)))
        self.n, self.total_bases = n.value, tot.value

    def lengths(self):
        out = np.empty(self.n, dtype=np.int64)
        check(lib().skr_fasta_lengths(self._h, out.ctypes.data_as(_p)))
        return out

    def headers(self):
        need = _i64(0)
        check(lib().skr_fasta_headers(self._h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(max(need.value, 1))
        check(lib().skr_fasta_headers(self._h, buf, need.value, None))
        text = buf.raw[:max(need.value - 1, 0)].decode("utf-8", "replace")  # (.raw: a header may hold a NUL byte)
        return text.split("\n") if text else []

    def pack(self, ctx, first=0, count=None, alphabet="AGTC"):
        count = self.n - first if count is None else count
        h = _p()
        check(lib().skr_fasta_pack(ctx._h, self._h, int(first), int(count), PackedSeqs._alpha(alphabet), C.byref(h)))
        return PackedSeqs(ctx, h)

    def free(self):
        if getattr(self, "_h", None) and not _shutdown:
            lib().skr_fasta_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class _OtherTable(dict):
    """str.translate table: letters of the alphabet keep a byte of their own, every other character shares one."""

    def __init__(self, mapping, other):
        super().__init__(mapping)
        self.other = other

    def __missing__(self, key):
        return self.other


def encode_text(seqs, alphabet):
    """Sequences and alphabet as bytes for the kernels, one CHARACTER -> one byte (the reference indexes `str`:
    kmer_counts.py:143-149), so len(seq) and every window position are kept.  Latin-1 text is taken as it stands.  With
    a character above U+00FF anywhere (a text-mode read of a UTF-8 file can produce one) each distinct letter of the
    alphabet gets a byte of its own and everything else one byte that no letter uses — what a window needs to be counted
    or skipped exactly as the dict look-up of the reference does."""
    joined = "".join(seqs)
    try:
        return joined.encode("latin-1"), alphabet.encode("latin-1")
    except UnicodeEncodeError:
        pass
    letters = list(dict.fromkeys(alphabet))
    if len(letters) > 255:
        raise NotImplementedError("an alphabet of more than 255 distinct characters is not supported")
    code = {ord(ch): chr(1 + i) for i, ch in enumerate(letters)}
    table = _OtherTable(code, chr(0))
    return joined.translate(table).encode("latin-1"), alphabet.translate(table).encode("latin-1")


class PackedSeqs:
    """Sequences packed 2 bits/base in HBM (skr_seqs)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self._h = handle
        n, tot, mx = _i64(0), _i64(0), _i64(0)
        check(lib().skr_seqs_info(handle, C.byref(n), C.byref(tot), C.byref(mx)))
        self.n, self.total_bases, self.max_len = n.value, tot.value, mx.value

    @staticmethod
    def _alpha(alphabet):
        if len(alphabet) != 4:
            raise NotImplementedError(
                "the MI355X counting path packs 2 bits per base and needs a 4-letter alphabet (got {!r})".format(alphabet))
        return alphabet if isinstance(alphabet, bytes) else alphabet.encode("latin-1")

    @classmethod
    def from_strings(cls, ctx, seqs, alphabet="AGTC"):
        seqs = list(seqs)
        lengths = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
        offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
        np.cumsum(lengths, out=offsets[1:])
        blob, alpha = encode_text(seqs, alphabet)  # 1 char -> 1 byte
        return cls.from_buffer(ctx, blob, offsets, alpha)

    @classmethod
    def from_buffer(cls, ctx, blob, offsets, alphabet="AGTC"):
        """`blob`: bytes / uint8 array of concatenated ASCII bases; `offsets`: int64 [n+1]."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if isinstance(blob, np.ndarray):
            blob = np.ascontiguousarray(blob, dtype=np.uint8)
            ptr = blob.ctypes.data_as(_p)
        else:
            ptr = C.cast(C.c_char_p(blob), _p)
        h = _p()
        check(lib().skr_seqs_pack(ctx._h, ptr, offsets.ctypes.data_as(_p), len(offsets) - 1, cls._alpha(alphabet),
                                  C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_fasta(cls, ctx, path, alphabet="AGTC"):
        h = _p()
        check(lib().skr_seqs_from_fasta(ctx._h, os.fsencode(path), cls._alpha(alphabet), C.byref(h)))
        return cls(ctx, h)

    def lengths(self):
        out = np.empty(self.n, dtype=np.int64)
        check(lib().skr_seqs_lengths(self._h, out.ctypes.data_as(_p)))
        return out

    def headers(self):
        need = _i64(0)
        check(lib().skr_seqs_headers(self._h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(max(need.value, 1))
        check(lib().skr_seqs_headers(self._h, buf, need.value, None))
        text = buf.raw[:max(need.value - 1, 0)].decode("utf-8", "replace")  # (.raw: a header may hold a NUL byte)
        return text.split("\n") if text else []

    def free(self):
        if getattr(self, "_h", None) and not _shutdown and getattr(self.ctx, "_h", None):
            lib().skr_seqs_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


# ----------------------------------------------------------------------------- operations --
def count_u32(ctx, seqs, k):
    out = ctx.empty(seqs.n, 4 ** k, np.uint32)
    check(lib().skr_count_u32(ctx._h, seqs._h, int(k), out._h))
    return out


def count_per_kb(ctx, seqs, k, log2_pre=False, dtype=np.float32, out=None):
    if out is None:
        out = ctx.empty(seqs.n, 4 ** k, dtype)
    check(lib().skr_count_per_kb(ctx._h, seqs._h, int(k), 1 if log2_pre else 0, out._h))
    return out


def count_generic(ctx, seqs, alphabet, k, dtype=np.float32, log2_pre=False):
    """Counts for an alphabet the 2-bit path does not cover (not 4 letters, or a repeated letter):
    [n, len(alphabet)^k] per-kb values (float32 / float64) or raw counts (uint32)."""
    seqs = list(seqs)
    lengths = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum(lengths, out=offsets[1:])
    blob, alpha = encode_text(seqs, alphabet)
    out = ctx.empty(len(seqs), len(alpha) ** k, dtype)
    check(lib().skr_count_generic(ctx._h, C.cast(C.c_char_p(blob), _p), offsets.ctypes.data_as(_p), len(seqs), alpha,
                                  len(alpha), int(k), 1 if log2_pre else 0, out._h))
    return out


class AsciiSeqs:
    """ASCII sequences resident on the device (skr_aseqs): count_generic_dev counts from them any number of times."""

    def __init__(self, ctx, blob, offsets):
        self.ctx = ctx
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.n = len(offsets) - 1
        self.total_bases = int(offsets[-1] - offsets[0]) if self.n > 0 else 0
        self._h = _p()
        buf = blob if isinstance(blob, (bytes, bytearray)) else np.ascontiguousarray(blob, dtype=np.uint8).tobytes()
        check(lib().skr_aseqs_create(ctx._h, C.cast(C.c_char_p(bytes(buf)), _p), offsets.ctypes.data_as(_p), self.n, C.byref(self._h)))

    def free(self):
        if getattr(self, "_h", None) and not _shutdown and getattr(self.ctx, "_h", None):
            lib().skr_aseqs_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


def count_generic_dev(ctx, aseqs, alphabet, k, dtype=np.float32, log2_pre=False, out=None):
    """count_generic from resident sequences (AsciiSeqs); `out`: an existing [n, len(alphabet)^k] matrix to fill."""
    alpha = alphabet.encode("latin-1", "replace")
    out = ctx.empty(aseqs.n, len(alpha) ** k, dtype) if out is None else out
    check(lib().skr_count_generic_dev(ctx._h, aseqs._h, alpha, len(alpha), int(k), 1 if log2_pre else 0, out._h))
    return out


# ---- BasicCounter's normalisation methods on a host matrix that is not float32 (skr_host_colstat / skr_host_apply)
NP_CODES = {np.dtype(np.float16): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int8): 3, np.dtype(np.int16): 4,
            np.dtype(np.int32): 5, np.dtype(np.int64): 6, np.dtype(np.uint8): 7, np.dtype(np.uint16): 8, np.dtype(np.uint32): 9,
            np.dtype(np.uint64): 10, np.dtype(np.bool_): 11}


def host_colstat(ctx, x, what):
    """np.mean (what = 'mean') / np.std ('std') along axis 0 of a C-contiguous host matrix of any supported dtype, evaluated
    on the device in numpy's order for that dtype; float16 in, float16 out — every other type gives float64."""
    out = np.empty(x.shape[1], dtype=np.float16 if x.dtype == np.float16 else np.float64)
    check(lib().skr_host_colstat(ctx._h, x.ctypes.data_as(_p), x.shape[0], x.shape[1], NP_CODES[x.dtype],
                                 {"mean": 0, "std": 1}[what], out.ctypes.data_as(_p)))
    return out


def column_major_like(a):
    """Does numpy reduce `a` along axis 0 COLUMN BY COLUMN (pairwise order) rather than row after row?  When axis 0 is the
    faster axis (Fortran order, or a strided view of one) or there is a single column; np.mean / np.std(axis=0) then add each
    column in the pairwise order of numpy's float loops."""
    return a.ndim == 2 and a.shape[0] >= 2 and (a.shape[1] == 1 or abs(a.strides[0]) < abs(a.strides[1]))


def host_colstat_colmajor(ctx, x, what):
    """np.mean / np.std along axis 0 of a host matrix that numpy reduces column by column (column_major_like): the pairwise
    order of its float loops, in pieces of the iterator's buffer, on the device.  float32 / float64 in their own arithmetic,
    float16 in numpy's half loops (float32 accumulators within a piece), integers and bool as the float64 values numpy casts
    them to piece by piece; returns a vector of the type np.mean / np.std give (x's own for floats, float64 otherwise)."""
    if x.dtype.kind != "f":
        x = x.astype(np.float64)  # exact for every integer numpy itself converts exactly; 'K' order: stays column-major
    xf = np.asfortranarray(x)
    out = np.empty(xf.shape[1], dtype=xf.dtype)
    check(lib().skr_host_colstat_colmajor(ctx._h, xf.ctypes.data_as(_p), xf.shape[0], xf.shape[1], NP_CODES[xf.dtype],
                                          {"mean": 0, "std": 1}[what], out.ctypes.data_as(_p)))
    return out


def host_apply(ctx, x, op, vec=None, out=None):
    """In place on the C-contiguous host matrix x: 'sub' / 'div' by the float32 or float64 vector `vec`, 'isub' by an int64
    vector, 'log2p1' (x += 1, out = log2(x)).  Returns has_nan."""
    code = {"sub": 0, "div": 1, "isub": 2, "log2p1": 3}[op]
    nan = _int(0)
    check(lib().skr_host_apply(ctx._h, x.ctypes.data_as(_p), x.shape[0], x.shape[1], NP_CODES[x.dtype], code,
                               vec.ctypes.data_as(_p) if vec is not None else None,
                               1 if vec is not None and vec.dtype == np.float64 else 0,
                               out.ctypes.data_as(_p) if out is not None else None, NP_CODES[out.dtype] if out is not None else 0,
                               C.byref(nan)))
    return bool(nan.value)


def colsum_seq(ctx, x, acc, center=None, center2=None, square=False):
    check(lib().skr_colsum_seq(ctx._h, x._h, _h(center), _h(center2), 1 if square else 0, acc._h))
    return acc


def colsum_seq_colmin(ctx, x, acc, colmin):
    """First pass of the column sums that also leaves the raw column minima in `colmin` [4, cols]."""
    check(lib().skr_colsum_seq_colmin(ctx._h, x._h, acc._h, colmin._h))
    return acc


class Chain:
    """The mailboxes of the column-sum chain across GPUs (skr_chain): peer stores instead of a send/recv per hop."""

    def __init__(self, ctx, cols_cap):
        self.ctx, self.cols_cap = ctx, int(cols_cap)
        self._h = _p()
        check(lib().skr_chain_create(ctx._h, self.cols_cap, C.byref(self._h)))

    def export(self):
        buf = C.create_string_buffer(64)
        check(lib().skr_chain_export(self._h, buf))
        return buf.raw

    def connect(self, rank, handles):
        """handles: the 64-byte handle of every rank, in rank order."""
        blob = b"".join(handles)
        assert len(blob) == 64 * len(handles)
        check(lib().skr_chain_connect(self._h, len(handles), int(rank), blob))

    def connect_local(self, rank, chains):
        arr = (_p * len(chains))(*[c._h for c in chains])
        check(lib().skr_chain_connect_local(self._h, len(chains), int(rank), arr))

    def colsum(self, x, acc, center=None, center2=None, square=False, colmin=None, defer_result=False):
        check(lib().skr_colsum_seq_chain(self._h, x._h, _h(center), _h(center2), 1 if square else 0, acc._h, _h(colmin),
                                         1 if defer_result else 0))
        return acc

    def result(self, acc):
        """The deferred half of colsum(defer_result=True): wait for the result box and copy it into acc."""
        check(lib().skr_chain_result(self._h, acc._h))
        return acc

    def timed_out(self):
        flag = _int(0)
        check(lib().skr_chain_check(self._h, C.byref(flag)))
        return bool(flag.value)

    def free(self):
        if getattr(self, "_h", None) and not _shutdown and getattr(self.ctx, "_h", None):
            lib().skr_chain_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


def vec_finish(ctx, v, n, take_sqrt=False):
    check(lib().skr_vec_finish(ctx._h, v._h, int(n), 1 if take_sqrt else 0))
    return v


def min_nan(ctx, x, center=None, scale=None):
    mn, nan = C.c_float(0), _int(0)
    check(lib().skr_min_nan(ctx._h, x._h, _h(center), _h(scale), C.byref(mn), C.byref(nan)))
    return np.float32(mn.value), bool(nan.value)


def apply(ctx, x, y=None, pre=False, center=None, scale=None, post=False, shift=0.0, want_nan=False):
    y = x if y is None else y
    nan = _int(0)
    check(lib().skr_apply(ctx._h, x._h, 1 if pre else 0, _h(center), _h(scale), 1 if post else 0,
                          C.c_float(shift), y._h, C.byref(nan) if want_nan else None))
    return y, bool(nan.value)


def normalize(ctx, x, log2, mean_mode, mean_vec, std_mode, std_vec):
    """In-place kmer_counts.py:201-209 on one GPU; returns (mean_out, std_out, has_nan)."""
    mean_out = ctx.empty(1, x.cols) if mean_mode == 1 else None
    std_out = ctx.empty(1, x.cols) if std_mode == 1 else None
    nan = _int(0)
    check(lib().skr_normalize(ctx._h, x._h, LOG2_CODES[log2], mean_mode, _h(mean_vec), std_mode, _h(std_vec),
                              _h(mean_out), _h(std_out), C.byref(nan)))
    return mean_out, std_out, bool(nan.value)


def row_standardize(ctx, x, z=None):
    z = ctx.empty(x.rows, x.cols, x.dtype) if z is None else z
    check(lib().skr_row_standardize(ctx._h, x._h, z._h))
    return z


def pearson_gemm(ctx, a, b, r, precision=PREC_FP32, symmetric=False, row0=0, col0=0):
    check(lib().skr_pearson_gemm(ctx._h, a._h, b._h, int(precision), 1 if symmetric else 0, r._h, int(row0), int(col0)))
    return r


def operand_fill(ctx, x, op=None, precision=PREC_F16X3, center=None, scale=None, post=False, shift=0.0, y=None,
                 row_standardize=True, want_nan=False):
    """Fused normalisation tail + row standardisation + operand layout; returns (op, has_nan)."""
    op = Operand(ctx, x.rows, x.cols, precision) if op is None else op
    nan = _int(0)
    check(lib().skr_operand_fill(ctx._h, x._h, _h(center), _h(scale), 1 if post else 0, C.c_float(shift), _h(y),
                                 1 if row_standardize else 0, op._h, C.byref(nan) if want_nan else None))
    return op, bool(nan.value)


def pearson_gemm_op(ctx, a, b, r, symmetric=False, row0=0, col0=0, lower=False):
    """lower: a plain block with the bits of the mirror of (b, a) — a block below the diagonal of a self-comparison
    computed from its own rows (skr_pearson_gemm_op, symmetric = 2)."""
    check(lib().skr_pearson_gemm_op(ctx._h, a._h, b._h, 2 if lower else (1 if symmetric else 0), r._h, int(row0), int(col0)))
    return r


def pearson_gemm_op_rows(ctx, a, full, a_row0, r, row0=0):
    """r[row0 + i, :] = rows a_row0 + i of the self-comparison of `full`, bit for bit (skr_pearson_gemm_op_rows)."""
    check(lib().skr_pearson_gemm_op_rows(ctx._h, a._h, full._h, int(a_row0), r._h, int(row0)))
    return r


def pearson_gemm_f64(ctx, a, b, r, K, symmetric=False, row0=0, col0=0):
    """float64 standardised (possibly zero-padded) rows: r[row0 + i, col0 + j] = <a_i, b_j> / K (skr_pearson_gemm_f64)."""
    check(lib().skr_pearson_gemm_f64(ctx._h, a._h, b._h, int(K), 1 if symmetric else 0, r._h, int(row0), int(col0)))
    return r


def pearson_gemm_op_mirror(ctx, a, b, r, row0, col0, rt, trow0, tcol0):
    """r[row0+i, col0+j] = <a_i, b_j>/K and rt[trow0+j, tcol0+i] = the same value."""
    check(lib().skr_pearson_gemm_op_mirror(ctx._h, a._h, b._h, r._h, int(row0), int(col0), rt._h, int(trow0), int(tcol0)))
    return r


def pearson(ctx, c1, c2, row_standardize=True, precision=PREC_FP32, r=None):
    r = ctx.empty(c1.rows, c2.rows, c1.dtype) if r is None else r
    check(lib().skr_pearson(ctx._h, c1._h, c2._h, 1 if row_standardize else 0, int(precision), r._h))
    return r


# ----------------------------------------------------------------------------- writers -----
FMT_FIXED6, FMT_SCI18, FMT_REPR = 0, 1, 2  # "%1.6f" (kmer_counts.py:241) / numpy's default "%.18e" / str(value)
_NP_DTYPES = {np.dtype(np.float32): F32, np.dtype(np.float64): F64, np.dtype(np.uint32): U32}


def npy_path(path):
    """np.save appends '.npy' unless the name already ends with it (kmer_counts.py:234 relies on it)."""
    path = os.fspath(path)
    return path if path.endswith(".npy") else path + ".npy"


def npy_create(path, dtype, rows, cols):
    """Create `path` (as given: no '.npy' is appended here) with numpy's header for a [rows, cols] array and its final
    size; returns the byte offset of row 0 (skr_npy_create).  Stripes are then written with Matrix.write_rows_at."""
    off = _i64(0)
    check(lib().skr_npy_create(os.fsencode(path), _NP_DTYPES[np.dtype(dtype)], int(rows), int(cols), C.byref(off)))
    return off.value


def save_npy(path, a):
    """np.save(path, a) for a device Matrix or a float32 / float64 / uint32 numpy array (1-D or 2-D)."""
    path = npy_path(path).encode()
    if isinstance(a, Matrix):
        check(lib().skr_mat_save_npy(a.ctx._h, a._h, 0, path))
        return
    a = np.ascontiguousarray(a)
    if a.dtype not in _NP_DTYPES or a.ndim not in (1, 2):
        np.save(path.decode(), a)
        return
    rows, cols = (1, a.shape[0]) if a.ndim == 1 else a.shape
    check(lib().skr_host_save_npy(a.ctypes.data_as(_p), _NP_DTYPES[a.dtype], rows, cols, 1 if a.ndim == 1 else 0, path))


def save_csv(path, a, fmt=FMT_FIXED6, threads=0):
    """np.savetxt(path, a, delimiter=',', fmt='%1.6f' | '%.18e') for a device Matrix or a 2-D float array."""
    path = os.fspath(path).encode()
    if isinstance(a, Matrix):
        check(lib().skr_mat_save_csv(a.ctx._h, a._h, int(fmt), int(threads), path))
        return
    a = np.ascontiguousarray(a)
    if a.dtype not in (np.float32, np.float64) or a.ndim != 2:
        np.savetxt(path.decode(), a, delimiter=",", fmt="%1.6f" if fmt == FMT_FIXED6 else "%.18e")
        return
    check(lib().skr_host_save_csv(a.ctypes.data_as(_p), _NP_DTYPES[a.dtype], a.shape[0], a.shape[1], int(fmt),
                                  int(threads), path))


def _joined_labels(labels, n):
    """'\\n'-joined utf-8 labels, or None when they cannot be passed that way (then pandas writes)."""
    labels = [str(x) for x in labels]
    if len(labels) != n or any("\n" in x or "\r" in x or "\x00" in x for x in labels):
        return None
    return "\n".join(labels).encode("utf-8")


def save_csv_labelled(path, a, index, columns, threads=0):
    """DataFrame(a, index=index, columns=columns).to_csv(path) (kmer_counts.py:236-240) for a device
    Matrix or a 2-D float32 / float64 array; anything the native writer does not cover goes to pandas."""
    rows, cols = (a.rows, a.cols) if isinstance(a, Matrix) else np.shape(a)
    rl, cl = _joined_labels(index, rows), _joined_labels(columns, cols)
    native = rl is not None and cl is not None and rows > 0 and cols > 0
    if native and isinstance(a, Matrix) and a.dtype != np.uint32:
        check(lib().skr_mat_save_csv_labelled(a.ctx._h, a._h, rl, cl, int(threads), os.fspath(path).encode()))
        return
    if isinstance(a, Matrix):
        a = a.to_numpy()
    arr = np.ascontiguousarray(a)
    if native and arr.ndim == 2 and arr.dtype in (np.float32, np.float64):
        check(lib().skr_host_save_csv_labelled(arr.ctypes.data_as(_p), _NP_DTYPES[arr.dtype], rows, cols, rl, cl,
                                               int(threads), os.fspath(path).encode()))
        return
    from pandas import DataFrame
    DataFrame(data=a, index=index, columns=columns).to_csv(path)


def _looks_numeric(label):
    try:
        float(label)
        return True
    except ValueError:
        return label.strip() == ""


def load_csv_labelled(path, threads=0):
    """pd.read_csv(path, index_col=0) (console_scripts.py:626-631) for the labelled count files this
    package and the reference write: returns (values float64 [rows, cols], row_labels, col_labels),
    or None when the file is outside the subset the native reader reproduces bit for bit (then the
    caller uses pandas): see csv_read.hip."""
    handle = _p()
    rc = lib().skr_csv_read(os.fspath(path).encode(), int(threads), C.byref(handle))
    if rc == -4:  # SKR_ERR_UNSUPPORTED
        return None
    check(rc)
    try:
        rows, cols = _i64(0), _i64(0)
        check(lib().skr_csv_shape(handle, C.byref(rows), C.byref(cols)))
        labels = []
        for which in (0, 1):
            need = _i64(0)
            check(lib().skr_csv_labels(handle, which, None, 0, C.byref(need)))
            buf = C.create_string_buffer(max(1, need.value))
            check(lib().skr_csv_labels(handle, which, buf, need.value, C.byref(need)))
            text = buf.raw[:need.value].decode("utf-8")
            labels.append(text.split("\n")[:-1] if need.value else [])
        # pandas would turn an all-numeric index into numbers and rename duplicate columns: not handled here
        if (labels[0] and all(_looks_numeric(x) for x in labels[0])) or len(set(labels[1])) != len(labels[1]):
            return None
        values = np.empty((rows.value, cols.value), dtype=np.float64)
        check(lib().skr_csv_values(handle, values.ctypes.data_as(_p)))
        return values, labels[0], labels[1]
    finally:
        lib().skr_csv_free(handle)


# ----------------------------------------------------------------------------- peer copies --
class Event:
    """A point in a ctx's compute or communication stream that any ctx of this process can wait for (skr_event)."""

    def __init__(self, ctx, comm=False):
        self._h = _p()
        check(lib().skr_event_record(ctx._h, 1 if comm else 0, C.byref(self._h)))

    def wait_on(self, ctx, comm=False):
        """ctx's compute (or communication) stream waits for this event."""
        check(lib().skr_event_wait(ctx._h, 1 if comm else 0, self._h))

    def free(self):
        if getattr(self, "_h", None) and not _shutdown:
            lib().skr_event_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


def peer_copy_rows(dst, drow0, src, srow0, nrows):
    """Rows of `src` (a Matrix of any Context of this process) -> rows of `dst`, on dst's ctx's communication stream."""
    check(lib().skr_peer_copy_rows(dst._h, int(drow0), src._h, int(srow0), int(nrows)))


# ----------------------------------------------------------------------------- RCCL --------
def comm_unique_id():
    buf = C.create_string_buffer(128)
    check(lib().skr_comm_unique_id(buf))
    return buf.raw


def comm_init(ctx, nranks, rank, uid):
    check(lib().skr_comm_init(ctx._h, int(nranks), int(rank), uid))


def comm_sendrecv(ctx, src, srow0, snrows, dst_rank, dst, drow0, dnrows, src_rank, want_ticket=True):
    """want_ticket=False: fire and forget (ordered on the communication stream only); returns None."""
    ticket = _i64(-1)
    check(lib().skr_comm_sendrecv(ctx._h, _h(src), int(srow0), int(snrows), int(dst_rank), _h(dst), int(drow0),
                                  int(dnrows), int(src_rank), C.byref(ticket) if want_ticket else None))
    return ticket.value if want_ticket else None


def comm_exchange(ctx, items):
    """items: [(src, srow0, snrows, dst_rank, dst, drow0, dnrows, src_rank)] posted as ONE grouped RCCL operation
    (transfers with different peers use their own xGMI links at once); returns one ticket."""
    n = len(items)
    mats_s = (_p * n)(*[_h(it[0]) for it in items])
    mats_d = (_p * n)(*[_h(it[4]) for it in items])
    i64 = lambda col: (C.c_int64 * n)(*[int(it[col]) for it in items])  # noqa: E731
    i32 = lambda col: (C.c_int * n)(*[int(it[col]) for it in items])    # noqa: E731
    ticket = _i64(-1)
    check(lib().skr_comm_exchange(ctx._h, n, mats_s, i64(1), i64(2), i32(3), mats_d, i64(5), i64(6), i32(7), C.byref(ticket)))
    return ticket.value


def comm_allgather_rows(ctx, shard, full, bounds):
    arr = (C.c_int64 * len(bounds))(*[int(b) for b in bounds])
    ticket = _i64(-1)
    check(lib().skr_comm_allgather_rows(ctx._h, shard._h, full._h, arr, C.byref(ticket)))
    return ticket.value


def comm_wait(ctx, ticket):
    check(lib().skr_comm_wait(ctx._h, int(ticket)))


def comm_allreduce(ctx, values, op="sum"):
    arr = (C.c_double * len(values))(*values)
    check(lib().skr_comm_allreduce_f64(ctx._h, arr, len(values), {"sum": 0, "max": 1, "min": 2}[op]))
    return list(arr)


def comm_barrier(ctx):
    check(lib().skr_comm_barrier(ctx._h))
