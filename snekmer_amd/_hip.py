"""_hip: ctypes binding of libsnekmer_hip.so (include/snekmer_hip.h).

There is deliberately no CPU fallback: importing this module never fails, but the first use
of the device path raises :class:`HipUnavailable` when the shared library or a GPU is
missing.  Host-side formatting (strings, numpy containers) lives in the callers; every
numeric result comes from the HIP kernels.
"""
import ctypes as C
import os
import sys
import threading
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SNEKMER_HIP_LIB: another build of the same library (tools/: A/B of compile-time kernel shapes in one GPU-box run)
LIB_PATH = os.environ.get("SNEKMER_HIP_LIB") or os.path.join(_HERE, "libsnekmer_hip.so")

SKM_OK = 0
ABI_VERSION = 7  # SKM_ABI_VERSION of include/snekmer_hip.h
ERRORS = {-1: "BADARG", -2: "NOMEM", -3: "HIP", -4: "OVERFLOW", -5: "UNSUPPORTED", -6: "COMM", -7: "STALE"}
COMM_ID_BYTES = 128
EVENT_SLOTS = 8  # SKM_EVENT_SLOTS


class P2POp(C.Structure):
    """skm_p2p_op of include/snekmer_hip.h."""

    _fields_ = [("peer", C.c_int32), ("array", C.c_int32), ("send_off", C.c_int64), ("send_bytes", C.c_int64),
                ("recv_off", C.c_int64), ("recv_bytes", C.c_int64)]


def under_profiler() -> bool:
    """True when a rocprofiler tool library is loaded into this process (rocprofv3 ... -- python3 ...).  Replaying a HIP graph
    under rocprofv3 --kernel-trace crashed inside the tool on ROCm 7.2 (hipGraphLaunch -> segmentation fault in the
    profiler's callback), so engine.Pipeline does not capture or replay graphs there."""
    return any("rocprof" in os.environ.get(v, "") for v in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))


class HipUnavailable(RuntimeError):
    """The HIP extension (or a GPU) is not available; there is no CPU fallback."""


class HipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libsnekmer_hip: SKM_E_{ERRORS.get(code, code)}: {message}")
        self.code = code


_lib = None
_lib_lock = threading.Lock()

_p = C.c_void_p
_i64 = C.c_int64
_SIGNATURES = {
    "skm_abi_version": (C.c_int, []),
    "skm_last_error": (C.c_char_p, []),
    "skm_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "skm_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "skm_get_option": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int]),
    "skm_create": (C.c_int, [C.c_int, C.POINTER(_p)]),
    "skm_create_confined": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_p)]),
    "skm_destroy": (C.c_int, [_p]),
    "skm_sync": (C.c_int, [_p]),
    "skm_graph_begin": (C.c_int, [_p]),
    "skm_graph_end": (C.c_int, [_p, C.POINTER(_p)]),
    "skm_graph_launch": (C.c_int, [_p, _p]),
    "skm_graph_nodes": (C.c_int, [_p, C.POINTER(_i64)]),
    "skm_graph_destroy": (C.c_int, [_p, _p]),
    "skm_event_record": (C.c_int, [_p, C.c_int]),
    "skm_stream_wait": (C.c_int, [_p, _p, C.c_int]),
    "skm_event_query": (C.c_int, [_p, C.c_int, C.POINTER(C.c_int)]),
    "skm_device_info": (C.c_int, [_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(_i64)]),
    "skm_malloc": (C.c_int, [_p, C.c_size_t, C.POINTER(_p)]),
    "skm_free": (C.c_int, [_p, _p]),
    "skm_mem_trim": (C.c_int, [_p, C.POINTER(_i64)]),
    "skm_mem_stats": (C.c_int, [_p, C.POINTER(_i64)]),
    "skm_debug_report": (C.c_int, [C.c_char_p, C.c_int]),
    "skm_host_alloc": (C.c_int, [_p, C.c_size_t, C.POINTER(_p)]),
    "skm_host_free": (C.c_int, [_p, _p]),
    "skm_memcpy_h2d": (C.c_int, [_p, _p, _p, C.c_size_t]),
    "skm_memcpy_h2d_async": (C.c_int, [_p, _p, _p, C.c_size_t]),
    "skm_memcpy_d2h": (C.c_int, [_p, _p, _p, C.c_size_t]),
    "skm_memcpy_d2d": (C.c_int, [_p, _p, _p, C.c_size_t]),
    "skm_memset": (C.c_int, [_p, _p, C.c_int, C.c_size_t]),
    "skm_profile_enable": (C.c_int, [_p, C.c_int]),
    "skm_profile_reset": (C.c_int, [_p]),
    "skm_profile_read": (C.c_int, [_p, C.c_char_p, C.POINTER(_i64), C.POINTER(C.c_double)]),
    "skm_profile_dump": (C.c_int, [_p, C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    "skm_recode": (C.c_int, [_p, _p, _p, _p, _i64, _p, _p]),
    "skm_kmer_codes": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, _p, _p, _i64, _p, _p]),
    "skm_count_csr": (
        C.c_int,
        [_p, _p, C.c_int, C.c_int, C.c_int, _p, _p, _i64, _i64, _i64, _i64, _p, _p, _p, _p, C.POINTER(_i64)],
    ),
    "skm_basis_build": (
        C.c_int,
        [_p, C.c_int, C.c_int, C.c_int, _i64, _i64, _p, _p, _p, _p, C.POINTER(_i64), _p, _p, _p, _p, _p, _p, _p, _p, _p],
    ),
    "skm_vectorize_csr": (
        C.c_int,
        [_p, _p, C.c_int, C.c_int, C.c_int, _p, _p, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    ),
    "skm_csr_transpose": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _p, _p, _p]),
    "skm_csr_concat_rowptr": (C.c_int, [_p, C.c_int, _p, _p, _p, _p]),
    "skm_csr_to_dense": (C.c_int, [_p, _i64, _p, _p, _p, _p, _i64, C.c_int, C.c_int, _p, _i64]),
    "skm_gather_columns": (C.c_int, [_p, _i64, _i64, C.c_int, _p, _i64, _p, _p]),
    "skm_narrow_u32_u8": (C.c_int, [_p, _i64, _p, _p]),
    "skm_widen_u8_u32": (C.c_int, [_p, _i64, _p, _p]),
    "skm_widen_i8_u32": (C.c_int, [_p, _i64, _p, _p]),
    "skm_csr_max_count": (C.c_int, [_p, _i64, _p, C.POINTER(C.c_uint32)]),
    "skm_csr_max_count_dev": (C.c_int, [_p, _i64, _i64, _p, _p, _p]),
    "skm_row_norms_csr": (C.c_int, [_p, _i64, _p, _p, _p, _p]),
    "skm_cosine_csr": (
        C.c_int,
        [_p, _i64, _p, _p, _p, _p, _i64, _i64, _p, _p, C.c_int, _p, _p, _i64, _i64, C.c_int, _p, _i64],
    ),
    "skm_cosine_csr_phase": (
        C.c_int,
        [_p, _p, C.c_int, _i64, _p, _p, _p, _p, _i64, _i64, _p, _p, C.c_int, _p, _p, _i64, _i64, C.c_int, _p, _i64],
    ),
    "skm_hamming_similarity_from_gram": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _p, _i64]),
    "skm_cosine_csr_stats": (C.c_int, [_p, _p]),
    "skm_heavy_panel_stats": (C.c_int, [_p, _p]),
    "skm_setsim_f64": (C.c_int, [_p, C.c_int, _i64, _i64, _i64, _p, _p, _p, _p, _i64, _p, _i64]),
    "skm_pairwise_f64": (C.c_int, [_p, C.c_int, C.c_double, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64]),
    "skm_row_top2": (C.c_int, [_p, _i64, _i64, _p, _i64, _p, _p]),
    "skm_csr_group_sum": (C.c_int, [_p, _i64, _i64, _p, _p, _p, _p, _i64, _p, _p, _p, C.POINTER(_i64)]),
    "skm_group_postings": (C.c_int, [_p, _i64, _i64, _p, _p, _i64, _p, _i64, _p, _p, _p, _p, C.POINTER(_i64)]),
    "skm_postings_to_csr": (C.c_int, [_p, _i64, _i64, _p, _p, _i64, _p, _p, _p]),
    "skm_gram_neighbors": (C.c_int, [_p, _i64, _p, _p, _p, _i64, _i64, _p, _p, C.c_int, _p, _p, _p, _i64, _i64, _i64, _p, _p, _p, C.POINTER(_i64), C.POINTER(_i64)]),
    "skm_neighbors_topk": (C.c_int, [_p, _i64, _i64, _p, _p, _p, _p, _p, _i64, C.c_int, C.c_int, _p, _p]),
    "skm_jaccard_distance_from_gram": (C.c_int, [_p, _i64, _i64, _p, _p, _p, _i64]),
    "skm_pair_work": (C.c_int, [_p, _i64, _p, C.POINTER(C.c_uint64)]),
    "skm_count_dense": (C.c_int, [_p, _p, C.c_int, C.c_int, _p, _p, _i64, C.c_int, _p, _i64]),
    "skm_cosine_dense_i8": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _p, _p, C.c_int, _p, _i64]),
    "skm_dense_to_csr": (C.c_int, [_p, _i64, _i64, C.c_int, _p, _i64, _i64, _p, _p, _p, C.POINTER(_i64)]),
    "skm_row_norms_i8": (C.c_int, [_p, _i64, _i64, _p, _p, _p]),
    "skm_cosine_dense_f64": (C.c_int, [_p, _i64, _i64, _i64, _p, _i64, _p, _i64, C.c_int, _p, _i64]),
    "skm_matrix_row_stats": (C.c_int, [_p, _i64, _i64, _p, _i64, _p, _p]),
    "skm_apply_top2": (C.c_int, [_p, _i64, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _i64, _i64, _p, _p, _p, _p]),
    "skm_fasta_index": (C.c_int, [_p, _i64, C.c_int, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(C.c_int)]),
    "skm_fasta_parse": (C.c_int, [_p, _i64, C.c_int, _i64, _i64, _p, _p, _p, _p]),
    "skm_csr_to_dense_i8": (C.c_int, [_p, _i64, _p, _p, _p, _i64, _p, _p, _p]),
    "skm_cosine_fixup_rows": (C.c_int, [_p, _i64, _p, _p, _p, _p, _i64, _i64, _p, _p, C.c_int, _p, _i64]),
    "skm_npz_write": (C.c_int, [C.c_char_p, C.c_int, _p, _p, _p, _p, _p, C.c_int, C.c_int, C.POINTER(_i64)]),
    "skm_rows_to_utf32": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _p]),
    "skm_decode_kmers_utf32": (C.c_int, [_p, C.c_int, C.c_int, C.c_int, _p, _p, _p, _i64, _p]),
    "skm_csr_remap_columns": (C.c_int, [_p, _i64, _p, _p, _p, _p, _i64, _p, _p, _p, C.POINTER(_i64)]),
    "skm_basis_select": (C.c_int, [_p, _i64, _p, _p, C.c_uint64, _p, _p, C.POINTER(_i64)]),
    "skm_comm_unique_id": (C.c_int, [_p]),
    "skm_comm_init": (C.c_int, [_p, C.c_int, C.c_int, _p]),
    "skm_comm_destroy": (C.c_int, [_p]),
    "skm_allgatherv": (C.c_int, [_p, _p, _p, _p]),
    "skm_alltoallv": (C.c_int, [_p, _p, _p, _p, _p]),
    "skm_plan_alltoallv": (C.c_int, [C.c_int, C.c_int, _p, _p, _p, _p]),
    "skm_alltoallv_multi": (C.c_int, [_p, C.c_int, _p, _p, _p, _p, _p]),
    "skm_plan_allgatherv": (C.c_int, [C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "skm_allgatherv_multi": (C.c_int, [_p, C.c_int, _p, _p, _p, _p]),
    "skm_bucket_partition": (C.c_int, [_p, C.c_int, C.c_int, _i64, _i64, _p, _p, _p, _i64, _p, _p, _p, _p, _p]),
    "skm_bucket_table_capacity": (_i64, [_i64]),
    "skm_bucket_postings": (C.c_int, [_p, C.c_int, C.c_int, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "skm_colidx_from_owners": (C.c_int, [_p, C.c_int, _i64, _p, _p, _p, _p, _p]),
    "skm_concat_colptr": (C.c_int, [_p, C.c_int, _p, _p, _p, _p]),
    "skm_colidx_lookup": (C.c_int, [_p, C.c_int, C.c_int, _i64, _p, _p, _p, _p, _p, _p]),
    "skm_embed_rowptr": (C.c_int, [_p, _i64, _i64, _i64, _p, _p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load_library():
    """dlopen the in-tree shared library and set argtypes; raises HipUnavailable if absent."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipUnavailable(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C snekmer_amd/csrc`. snekmer_amd has no CPU fallback."
            )
        # HIP spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (4 unless set) and two streams on one
        # queue run one after the other: with five streams alive engine.OverlappedPipeline's two fell on the same queue and
        # a step took 11.9 instead of 9.7 ms (tools/ab_overlapped.py, SKM_AB_EXTRA_CTX).  The runtime reads the variable
        # when it starts, so it is set here, before the library (and with it the runtime) is loaded, unless the caller
        # chose a value; a process that initialised HIP earlier (another library) keeps what it had.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as exc:  # e.g. libamdhip64 not found
            raise HipUnavailable(f"cannot load {LIB_PATH}: {exc}") from exc
        lib.skm_abi_version.restype = C.c_int
        if lib.skm_abi_version() != ABI_VERSION:  # e.g. a stale build named by SNEKMER_HIP_LIB
            raise HipUnavailable(f"{LIB_PATH} has ABI version {lib.skm_abi_version()}, this binding needs {ABI_VERSION}: rebuild it")
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def _check(lib, status: int):
    if status != SKM_OK:
        raise HipError(status, lib.skm_last_error().decode("utf-8", "replace"))


FREE_ERRORS = []  # messages of skm_free calls that failed (SKM_GUARD=1: an array was overrun); tests assert it stays empty


def debug_report() -> str:
    """skm_debug_report: streams idle / busy per context, timed kernels that started and did not finish, pool counters.
    Safe to call from a watchdog thread while another thread is stuck inside a library call."""
    lib = load_library()
    buf = C.create_string_buffer(1 << 16)
    lib.skm_debug_report(buf, len(buf))
    return buf.value.decode("utf-8", "replace")


CALL_TRACE = None  # a collections.deque(maxlen=...) while a tool wants the last library calls recorded
CALL_SYNC = False

OPTION_NAMES = ("SKM_SORT", "SKM_COSINE_PATH", "SKM_HEAVY_PANEL", "SKM_HEAVY_PACK", "SKM_COSINE_OVERLAP", "SKM_GRAM_SHAPE", "SKM_DENSE_VARIANT")


def set_option(name: str, value) -> None:
    """skm_set_option: a process-wide switch between exact kernels (include/snekmer_hip.h lists them).  The library reads
    the environment variable of the same name once, when it is first used; afterwards this is the way to change one.
    value None: back to what the environment said.  Unknown names and values raise HipError (SKM_E_BADARG)."""
    lib = load_library()
    _check(lib, lib.skm_set_option(name.encode(), None if value is None else str(value).encode()))


def get_option(name: str):
    """Current value of an option as a string, None when unset."""
    lib = load_library()
    buf = C.create_string_buffer(64)
    _check(lib, lib.skm_get_option(name.encode(), buf, 64))
    return buf.value.decode() or None


class options:
    """`with _hip.options(SKM_SORT="rocprim"): ...`: set for the block, previous values back afterwards."""

    def __init__(self, **values):
        self.values, self.before = values, {}

    def __enter__(self):
        for name, value in self.values.items():
            self.before[name] = get_option(name)
            set_option(name, value)
        return self

    def __exit__(self, *exc):
        for name, value in self.before.items():
            set_option(name, None)  # the environment's value ...
            if value is not None and get_option(name) != value:
                set_option(name, value)  # ... or whatever an enclosing block had set
        return False


def device_count() -> int:
    lib = load_library()
    n = C.c_int(0)
    status = lib.skm_device_count(C.byref(n))
    if status != SKM_OK:
        return 0
    return n.value


class DeviceArray:
    """A typed 1-D/2-D view of device memory owned by a Context."""

    def __init__(self, ctx: "Context", shape, dtype):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape)) if self.shape else 1
        self.nbytes = self.size * self.dtype.itemsize
        self.ptr = ctx._malloc(max(self.nbytes, 1))

    def free(self):
        if self.ptr is not None and self.ctx is not None:
            owner = self.ctx
            if owner.handle is None:
                # the context that allocated the array was closed first (engine.OverlappedPipeline's buffer sets outlive side
                # contexts a caller closes): the memory is the device's, any live context of that device can release it
                owner = _default_ctx if (_default_ctx is not None and _default_ctx.handle is not None
                                         and _default_ctx.device == self.ctx.device) else None
            if owner is not None:
                owner._free(self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def upload(self, host: np.ndarray) -> "DeviceArray":
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.size != self.size:
            raise ValueError(f"upload size mismatch: {host.size} vs {self.size}")
        self.ctx._h2d(self.ptr, host)
        return self

    def download(self, count: Optional[int] = None, offset: int = 0) -> np.ndarray:
        """Copy `count` elements starting at element `offset` (flat) to a new host array."""
        if count is None:
            count = self.size - offset
            shape = self.shape if offset == 0 else (count,)
        else:
            shape = (count,)
        out = np.empty(count, dtype=self.dtype)
        if count:
            self.ctx._d2h(out, self.ptr + offset * self.dtype.itemsize)
        return out.reshape(shape)

    def at(self, offset_elems: int) -> int:
        return self.ptr + int(offset_elems) * self.dtype.itemsize


class Context:
    """One device + one HIP stream (skm_ctx).  Not re-entrant: one host thread at a time per context (the
    small-batch arena of engine.recode_host / kmer_codes_host is per context and unguarded); distinct contexts may be
    used from distinct threads."""

    def __init__(self, device: int = 0, cu_groups: Optional[Tuple[int, int]] = None):
        """`cu_groups=(first, last)`: the context's stream may only use those compute-unit groups of 0..7
        (skm_create_confined): a side context for work that should run beside another context's kernels."""
        self.lib = load_library()
        if device_count() <= device:
            raise HipUnavailable(
                f"no HIP device {device} visible ({device_count()} found); snekmer_amd has no CPU fallback"
            )
        handle = _p()
        if cu_groups is None:
            _check(self.lib, self.lib.skm_create(device, C.byref(handle)))
        else:
            _check(self.lib, self.lib.skm_create_confined(device, int(cu_groups[0]), int(cu_groups[1]), C.byref(handle)))
        self.cu_groups = cu_groups
        self.handle = handle
        self.device = device

    def close(self):
        """Free the context.  Pinned host memory handed out by host_alloc (the small-batch arena of engine.py
        included) is freed with it: numpy views a caller kept over it are dangling afterwards, and any later call on
        this context raises."""
        if getattr(self, "handle", None) is not None:
            self.__dict__.pop("_arena", None)  # its views point into the pinned memory freed below
            for ptr in self.__dict__.pop("_pinned", []):
                self.lib.skm_host_free(self.handle, _p(ptr))
            self.lib.skm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- memory
    def _malloc(self, nbytes: int) -> int:
        out = _p()
        _check(self.lib, self.lib.skm_malloc(self.handle, nbytes, C.byref(out)))
        return out.value

    def _free(self, ptr: int):
        # (called from __del__: no exception can leave it; a failure - SKM_GUARD's overrun report - is kept and printed)
        status = self.lib.skm_free(self.handle, _p(ptr))
        if status != SKM_OK:
            msg = self.lib.skm_last_error().decode("utf-8", "replace")
            FREE_ERRORS.append(msg)
            print(f"libsnekmer_hip: skm_free: {msg}", file=sys.stderr, flush=True)

    def mem_stats(self) -> dict:
        """skm_mem_stats: what the device's array pool holds and has done (include/snekmer_hip.h)."""
        out = (_i64 * 8)()
        _check(self.lib, self.lib.skm_mem_stats(self.handle, out))
        keys = ("live_bytes", "parked_bytes", "hipMalloc_calls", "hipFree_calls", "reused", "parked_blocks", "cached_streams", "busy_skipped")
        return dict(zip(keys, (int(v) for v in out)))

    def trim(self) -> int:
        """Give every parked block back to the runtime (waits for the device first); bytes released."""
        out = _i64(0)
        _check(self.lib, self.lib.skm_mem_trim(self.handle, C.byref(out)))
        return int(out.value)

    def h2d_async(self, dptr: int, pinned: np.ndarray, nbytes: Optional[int] = None):
        """Queue a copy from pinned host memory (host_alloc) on this context's stream without waiting; the source must
        stay untouched until the copy has run (record_event + event_done, or sync)."""
        _check(self.lib, self.lib.skm_memcpy_h2d_async(self.handle, _p(dptr), pinned.ctypes.data_as(_p),
                                                        pinned.nbytes if nbytes is None else int(nbytes)))

    def event_done(self, slot: int) -> bool:
        """True once everything queued before the last record_event(slot) has run (never waits)."""
        done = C.c_int(0)
        _check(self.lib, self.lib.skm_event_query(self.handle, slot, C.byref(done)))
        return bool(done.value)

    def host_alloc(self, nbytes: int) -> np.ndarray:
        """Pinned, device-addressable host bytes (skm_host_alloc) as a uint8 array; `arr.ctypes.data` is valid as a
        device pointer.  Lives as long as the context (freed in close())."""
        out = _p()
        _check(self.lib, self.lib.skm_host_alloc(self.handle, C.c_size_t(nbytes), C.byref(out)))
        self.__dict__.setdefault("_pinned", []).append(out.value)
        return np.ctypeslib.as_array(C.cast(out.value, C.POINTER(C.c_uint8)), shape=(max(nbytes, 1),))[:nbytes]

    def _h2d(self, dptr: int, host: np.ndarray):
        _check(self.lib, self.lib.skm_memcpy_h2d(self.handle, _p(dptr), host.ctypes.data_as(_p), host.nbytes))

    def _d2h(self, host: np.ndarray, dptr: int):
        _check(self.lib, self.lib.skm_memcpy_d2h(self.handle, host.ctypes.data_as(_p), _p(dptr), host.nbytes))

    def empty(self, shape, dtype) -> DeviceArray:
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype) -> DeviceArray:
        arr = DeviceArray(self, shape, dtype)
        _check(self.lib, self.lib.skm_memset(self.handle, _p(arr.ptr), 0, arr.nbytes))
        return arr

    def to_device(self, host: np.ndarray, dtype=None) -> DeviceArray:
        host = np.ascontiguousarray(host, dtype=dtype)
        return DeviceArray(self, host.shape, host.dtype).upload(host)

    def sync(self):
        """Wait for the context's stream.  Errors that only the device can detect are reported HERE (and by downloads), i.e.
        possibly calls later than their cause: a HipError SKM_E_BADARG from sync() / download() naming `max_seq_len` refers
        to an earlier count_csr / vectorize call on this context whose bound was smaller than one of its sequences (those
        rows were left empty); the word is sticky until reported once."""
        _check(self.lib, self.lib.skm_sync(self.handle))

    # -- HIP graphs (skm_graph_*): record a fixed sequence of calls once, replay it with one launch
    def graph_begin(self):
        _check(self.lib, self.lib.skm_graph_begin(self.handle))

    def graph_end(self) -> "Graph":
        g = _p()
        _check(self.lib, self.lib.skm_graph_end(self.handle, C.byref(g)))
        return Graph(self, g)

    def record_event(self, slot: int):
        """Mark the current end of this context's stream in `slot` (skm_event_record)."""
        _check(self.lib, self.lib.skm_event_record(self.handle, slot))

    def wait_event(self, src: "Context", slot: int):
        """Everything queued on this context from now on waits for the mark `src` last recorded in `slot`."""
        _check(self.lib, self.lib.skm_stream_wait(self.handle, src.handle, slot))

    def device_info(self) -> Tuple[str, int, int]:
        name = C.create_string_buffer(256)
        cus = C.c_int(0)
        mem = _i64(0)
        _check(self.lib, self.lib.skm_device_info(self.handle, name, 256, C.byref(cus), C.byref(mem)))
        return name.value.decode(), cus.value, mem.value

    # -- profiling
    def profile_enable(self, on: bool = True):
        _check(self.lib, self.lib.skm_profile_enable(self.handle, 1 if on else 0))
        self.profiling = bool(on)  # (engine.Pipeline runs eagerly while per-kernel timing is on: a graph replay records no events)

    def profile_reset(self):
        _check(self.lib, self.lib.skm_profile_reset(self.handle))

    def profile_read(self, prefix: str) -> Tuple[int, float]:
        n = _i64(0)
        ms = C.c_double(0.0)
        _check(self.lib, self.lib.skm_profile_read(self.handle, prefix.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def profile_dump(self) -> dict:
        need = C.c_int(0)
        _check(self.lib, self.lib.skm_profile_dump(self.handle, None, 0, C.byref(need)))
        buf = C.create_string_buffer(max(need.value, 1))
        _check(self.lib, self.lib.skm_profile_dump(self.handle, buf, need.value, C.byref(need)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split("\t")
            out[name] = (int(cnt), float(ms))
        return out

    # -- raw entry points (thin; shapes are the callers' business)
    def call(self, name: str, *args):
        if self.handle is None:
            raise HipError(-1, f"{name}: the context is closed")
        if CALL_TRACE is not None:  # (tools/fuzz_watch.py: the last library calls, for a run that stops answering)
            CALL_TRACE.append((name, id(self), tuple(a.value if hasattr(a, "value") and isinstance(a.value, int) else None for a in args)))
        _check(self.lib, getattr(self.lib, name)(self.handle, *args))
        if CALL_SYNC:  # (the same tool, second run: wait for the device after every call so that a hang names its call)
            _check(self.lib, self.lib.skm_sync(self.handle))


class Graph:
    """A captured sequence of calls on one context (skm_graph_begin .. skm_graph_end)."""

    def __init__(self, ctx: Context, handle):
        self.ctx, self.handle = ctx, handle

    def launch(self):
        _check(self.ctx.lib, self.ctx.lib.skm_graph_launch(self.ctx.handle, self.handle))

    @property
    def nodes(self) -> int:
        """Kernels, fills and copies in the capture: the GPU operations one replay (= one eager run of the same calls) issues."""
        n = _i64(0)
        _check(self.ctx.lib, self.ctx.lib.skm_graph_nodes(self.handle, C.byref(n)))
        return int(n.value)

    def close(self):
        # (a context that was closed first has destroyed its live graphs itself: skm_destroy)
        if self.handle is not None and self.ctx.handle is not None:
            self.ctx.lib.skm_graph_destroy(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    """Process-wide context on the device named by SNEKMER_DEVICE / LOCAL_RANK (default 0)."""
    global _default_ctx
    if _default_ctx is None:
        dev = int(os.environ.get("SNEKMER_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _default_ctx = Context(dev)
    return _default_ctx
