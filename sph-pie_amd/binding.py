"""ctypes binding of include/pie_scan.h (libpie_hip.so).  Thin: every method is one C-ABI call.

There is no CPU fallback here: if the HIP library is missing this raises, and if no GPU is present
`PieScan()` raises with the library's own error text."""
import ctypes as C
import os

import numpy as np

from .build import HIP_LIB

PIE_GEN_INTERVAL = 1
PIE_GEN_CLUSTERED = 2
PIE_GEN_TIME_ORDERED = 4
PIE_END_NONE = -(2 ** 63)
INT64_MIN = -(2 ** 63)

PIE_E_CAPACITY = -5


class PieError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("pie_scan error %d: %s" % (code, text))
        self.code = code


class PieTableInfo(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("has_keys", C.c_uint32), ("rows", C.c_uint64), ("users", C.c_uint64),
        ("table_bytes", C.c_uint64), ("derived_bytes", C.c_uint64), ("workspace_bytes", C.c_uint64),
        ("index_build_ms", C.c_double), ("ordered_rows", C.c_uint64), ("ordered_bytes", C.c_uint64),
        ("ordered_build_ms", C.c_double), ("ordered_builds", C.c_uint64), ("ordered_positions", C.c_uint64),
        ("ordered_respreads", C.c_uint64),
    ]


class PieQuery(C.Structure):
    _fields_ = [("now", C.c_int64), ("cutoff", C.c_int64), ("mask", C.c_uint64)]


PIE_BATCH_MAX = 64


class PieStats(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_profiled", C.c_uint32), ("rows", C.c_uint64), ("users", C.c_uint64),
        ("selected", C.c_uint64), ("alg_bytes", C.c_uint64), ("k1_ms_sum", C.c_double), ("scan_ms_sum", C.c_double),
        ("max_bucket", C.c_uint32), ("n_segments", C.c_uint32), ("n_big", C.c_uint32), ("k1_blocks", C.c_uint32),
        ("k1_variant", C.c_uint32), ("key_ambiguous", C.c_uint32), ("live", C.c_uint64), ("candidates", C.c_uint64),
    ]


# every symbol include/pie_scan.h declares: (name, restype, argtypes)
_P = C.c_void_p
_SIGS = [
    ("pie_abi_version", C.c_int, []),
    ("pie_device_count", C.c_int, []),
    ("pie_ctx_create", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("pie_ctx_destroy", C.c_int, [_P]),
    ("pie_last_error", C.c_char_p, [_P]),
    ("pie_ctx_set_stream", C.c_int, [_P, _P]),
    ("pie_ctx_aux_stream", C.c_int, [_P, C.POINTER(_P)]),
    ("pie_load_columns", C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_int32]),
    ("pie_gen_synthetic", C.c_int, [_P, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_uint32]),
    ("pie_gen_synthetic_cdf", C.c_int, [_P, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_uint32, _P]),
    ("pie_save_columns", C.c_int, [_P, C.c_char_p]),
    ("pie_load_columns_dir", C.c_int, [_P, C.c_char_p]),
    ("pie_read_columns", C.c_int, [_P, _P, _P, _P, _P, C.c_size_t]),
    ("pie_set_end", C.c_int, [_P, _P, _P, C.c_size_t]),
    ("pie_delete_user", C.c_int, [_P, C.c_int32, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("pie_prune_before", C.c_int, [_P, C.c_int64, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("pie_retention_purge", C.c_int, [_P, C.c_int64, C.c_int32, C.c_int64, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("pie_retention_purge_tz", C.c_int, [_P, C.c_int64, C.c_int32, _P, _P, C.c_int32, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("pie_append_rows", C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_int32]),
    ("pie_set_disciplines", C.c_int, [_P, C.c_uint64, C.c_int32]),
    ("pie_scan", C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("pie_scan_device", C.c_int, [_P, C.c_int64, C.c_int64, C.POINTER(C.c_size_t)]),
    ("pie_scan_begin", C.c_int, [_P, C.c_int64, C.c_int64]),
    ("pie_scan_finish", C.c_int, [_P, C.POINTER(C.c_size_t)]),
    ("pie_scan_begin_packed", C.c_int, [_P, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.c_size_t]),
    ("pie_scan_finish_packed", C.c_int, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    ("pie_scan_begin_packed2", C.c_int, [_P, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    ("pie_host_alloc", C.c_int, [_P, C.c_size_t, C.POINTER(_P), C.POINTER(_P)]),
    ("pie_host_free", C.c_int, [_P, _P]),
    ("pie_set_scan_form", C.c_int, [_P, C.c_int]),
    ("pie_set_ordered_run", C.c_int, [_P, C.c_int]),
    ("pie_set_batch_lanes", C.c_int, [_P, C.c_int]),
    ("pie_batch_lanes", C.c_int, [_P]),
    ("pie_batch_room", C.c_int, [_P]),
    ("pie_scan_batch_flush", C.c_int, [_P]),
    ("pie_batch_pack_union_device", C.c_int, [_P, _P, C.c_size_t, C.c_size_t]),
    ("pie_table_info_get", C.c_int, [_P, C.POINTER(PieTableInfo)]),
    ("pie_scan_batch_begin", C.c_int, [_P, C.POINTER(PieQuery), C.c_int]),
    ("pie_scan_batch_finish", C.c_int, [_P, C.POINTER(C.c_size_t)]),
    ("pie_scan_batch", C.c_int, [_P, C.POINTER(PieQuery), C.c_int, C.POINTER(C.c_size_t)]),
    ("pie_scan_batch_begin_union", C.c_int, [_P, C.POINTER(PieQuery), C.c_int, C.c_void_p, C.c_size_t, C.c_size_t]),
    ("pie_batch_union_device_ptrs", C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_size_t)]),
    ("pie_batch_read_union", C.c_int, [_P, _P, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("pie_scan_batch_begin_packed", C.c_int, [_P, C.POINTER(PieQuery), C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t]),
    ("pie_scan_batch_finish_packed", C.c_int, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    ("pie_batch_read_results", C.c_int, [_P, C.c_int, _P, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("pie_batch_result_device_ptrs", C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    ("pie_batch_read_user_feed", C.c_int, [_P, C.c_int, C.c_int32, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("pie_batch_fetch_requests", C.c_int, [_P, _P, _P, C.c_size_t, C.c_size_t, _P, _P, _P, _P, _P, C.POINTER(C.c_size_t)]),
    ("pie_read_results", C.c_int, [_P, _P, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("pie_read_user_feed", C.c_int, [_P, C.c_int32, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("pie_result_device_ptrs", C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    ("pie_copy_results_device", C.c_int, [_P, _P, _P, _P, C.c_size_t]),
    ("pie_pack_results_device", C.c_int, [_P, _P, C.c_size_t, C.c_size_t]),
    ("pie_fetch_rows", C.c_int, [_P, _P, C.c_size_t, _P, _P, _P, _P]),
    ("pie_expired_queue", C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("pie_archive_queue", C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("pie_archive_stats", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    ("pie_set_profiling", C.c_int, [_P, C.c_int]),
    ("pie_stats_get", C.c_int, [_P, C.POINTER(PieStats)]),
    ("pie_stats_reset", C.c_int, [_P]),
    ("pie_synchronize", C.c_int, [_P]),
    ("pie_shard_of", C.c_int32, [C.c_int32, C.c_int32]),
    ("pie_shard_table", C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_size_t), C.POINTER(C.c_int32)]),
    ("pie_shard_maps", C.c_int, [_P, _P, _P]),
    ("pie_comm_create", C.c_int, [C.POINTER(C.c_int32), C.c_int32, C.POINTER(_P)]),
    ("pie_comm_unique_id", C.c_int, [_P]),
    ("pie_comm_create_rank", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("pie_comm_destroy", C.c_int, [_P]),
    ("pie_comm_last_error", C.c_char_p, [_P]),
    ("pie_comm_world", C.c_int32, [_P]),
    ("pie_comm_local_ranks", C.c_int32, [_P]),
    ("pie_comm_ctx", _P, [_P, C.c_int32]),
    ("pie_comm_gen_synthetic_sharded", C.c_int, [_P, C.c_uint64, C.c_int64, C.c_int32, C.c_int32, C.c_uint32]),
    ("pie_comm_scan_batch_gather", C.c_int, [_P, C.POINTER(PieQuery), C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    ("pie_comm_reserve", C.c_int, [_P, C.c_int32, C.c_int32, C.c_size_t]),
    ("pie_comm_needed_cap", C.c_size_t, [_P]),
    ("pie_comm_step_reserve", C.c_int, [_P, C.c_int32, C.c_int32, C.c_size_t]),
    ("pie_comm_step_begin", C.c_int, [_P, C.POINTER(PieQuery), C.c_int32]),
    ("pie_comm_step_finish", C.c_int, [_P, C.POINTER(C.c_size_t)]),
    ("pie_comm_step_collect", C.c_int, [_P, C.POINTER(C.c_int64)]),
    ("pie_comm_step_gathered_ptr", C.c_int, [_P, C.c_int32, C.c_int64, C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("pie_comm_step_read_gathered", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int64, _P, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("pie_comm_gathered_device_ptr", C.c_int, [_P, C.c_int32, C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("pie_comm_read_gathered", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
]
ABI_SYMBOLS = [s[0] for s in _SIGS]

_lib = None


def load_library(path=None):
    """dlopen libpie_hip.so and type its symbols.  Raises if the library was not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("PIE_HIP_LIB") or HIP_LIB   # PIE_HIP_LIB: A/B builds of the same source (tuning runs)
    if not os.path.exists(path):
        raise RuntimeError("%s is missing — build it first: python -c 'import __graft_entry__ as g; g.build()' "
                           "(there is no CPU fallback for the scan path)" % path)
    lib = C.CDLL(path)
    for name, res, args in _SIGS:
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _col(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


class PieScan:
    """One context = one GPU = one process."""

    def __init__(self, device=0, lib_path=None):
        self._lib = load_library(lib_path)
        self._ctx = _P()
        rc = self._lib.pie_ctx_create(int(device), C.byref(self._ctx))
        if rc != 0:
            text = self._lib.pie_last_error(None).decode()
            self._ctx = None
            raise PieError(rc, text)
        self.n = 0
        self.n_users = 0
        self._begun = []   # scans begun and not finished, oldest first: True = begun with scan_begin_packed
        self._m_buf, self._ready_buf = (C.c_size_t * PIE_BATCH_MAX)(), C.c_int(0)
        self._ready_ref = C.byref(self._ready_buf)

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.pie_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise PieError(rc, self._lib.pie_last_error(self._ctx).decode())

    # ---- table
    def load_columns(self, start, end, user, disc, n_users):
        start, end = _col(start, np.int64), _col(end, np.int64)
        user, disc = _col(user, np.int32), _col(disc, np.int32)
        n = start.shape[0]
        if not (end.shape[0] == user.shape[0] == disc.shape[0] == n):
            raise ValueError("column lengths differ")
        self._check(self._lib.pie_load_columns(self._ctx, _ptr(start), _ptr(end), _ptr(user), _ptr(disc), n, int(n_users)))
        self.n, self.n_users = n, int(n_users)

    def gen_synthetic(self, seed, n_total, row0, n, n_users, n_disc, flags=0):
        self._check(self._lib.pie_gen_synthetic(self._ctx, seed, n_total, row0, n, n_users, n_disc, flags))
        self.n, self.n_users = int(n), int(n_users)

    def gen_synthetic_cdf(self, seed, n_total, row0, n, n_users, n_disc, flags, cdf):
        """Skewed users: cdf = n_users ascending uint64 thresholds (floor(CDF_k * 2^64))."""
        cdf = np.ascontiguousarray(cdf, np.uint64)
        if cdf.shape[0] != n_users:
            raise ValueError("cdf must have n_users entries")
        self._check(self._lib.pie_gen_synthetic_cdf(self._ctx, seed, n_total, row0, n, n_users, n_disc, flags, _ptr(cdf)))
        self.n, self.n_users = int(n), int(n_users)

    def save_columns(self, directory):
        self._check(self._lib.pie_save_columns(self._ctx, os.fsencode(directory)))

    def load_columns_dir(self, directory):
        self._check(self._lib.pie_load_columns_dir(self._ctx, os.fsencode(directory)))
        st = self.stats()
        self.n, self.n_users = int(st["rows"]), int(st["users"])

    def read_columns(self):
        n = self.n
        s, e = np.empty(n, np.int64), np.empty(n, np.int64)
        u, d = np.empty(n, np.int32), np.empty(n, np.int32)
        self._check(self._lib.pie_read_columns(self._ctx, _ptr(s), _ptr(e), _ptr(u), _ptr(d), n))
        return s, e, u, d

    def set_end(self, rows, new_end):
        rows, new_end = _col(rows, np.int32), _col(new_end, np.int64)
        self._check(self._lib.pie_set_end(self._ctx, _ptr(rows), _ptr(new_end), rows.shape[0]))

    def append_rows(self, start, end, user, disc, n_users):
        start, end = _col(start, np.int64), _col(end, np.int64)
        user, disc = _col(user, np.int32), _col(disc, np.int32)
        self._check(self._lib.pie_append_rows(self._ctx, _ptr(start), _ptr(end), _ptr(user), _ptr(disc), start.shape[0], int(n_users)))
        self.n += start.shape[0]
        self.n_users = int(n_users)

    def delete_user(self, user):
        """-> ascending row indices that were tombstoned."""
        k = C.c_size_t(0)
        rows = np.empty(max(self.n, 1), np.int32)
        self._check(self._lib.pie_delete_user(self._ctx, int(user), _ptr(rows), self.n, C.byref(k)))
        return rows[: k.value].copy()

    def prune_before(self, cutoff):
        """Tombstone rows with start < cutoff; -> their ascending row indices."""
        k = C.c_size_t(0)
        rows = np.empty(max(self.n, 1), np.int32)
        self._check(self._lib.pie_prune_before(self._ctx, int(cutoff), _ptr(rows), self.n, C.byref(k)))
        return rows[: k.value].copy()

    def retention_purge(self, now, months=2, tz_offset_ms=0, tz_table=None):
        """Tombstone rows with now >= addMonths(start, months) (JS calendar-month arithmetic on a LOCAL date); -> their ascending
        indices.  tz_table = (transitions_utc_ms, offsets_ms) for a zone with daylight saving (binding.tz_table(zone)); else the
        fixed offset tz_offset_ms."""
        k = C.c_size_t(0)
        rows = np.empty(max(self.n, 1), np.int32)
        if tz_table is not None:
            T, off = np.ascontiguousarray(tz_table[0], np.int64), np.ascontiguousarray(tz_table[1], np.int64)
            if off.shape[0] != T.shape[0] + 1:
                raise ValueError("a table of n transitions carries n + 1 offsets")
            self._check(self._lib.pie_retention_purge_tz(self._ctx, int(now), int(months), _ptr(T), _ptr(off), int(T.shape[0]), _ptr(rows), self.n, C.byref(k)))
        else:
            self._check(self._lib.pie_retention_purge(self._ctx, int(now), int(months), int(tz_offset_ms), _ptr(rows), self.n, C.byref(k)))
        return rows[: k.value].copy()

    def set_disciplines(self, mask, n_disc):
        self._check(self._lib.pie_set_disciplines(self._ctx, mask & (2 ** 64 - 1), int(n_disc)))

    # ---- scan
    def scan(self, now, cutoff, idx_cap=None):
        """-> (counts[U] int32, offsets[U+1] int64, idx[M] int32), host arrays."""
        U = self.n_users
        if idx_cap is None:
            # size the row list by what the scan selected (a table-sized buffer per call would cost more than the scan)
            self.scan_device(now, cutoff)
            return self.read_results()
        counts, offsets = np.empty(U, np.int32), np.empty(U + 1, np.int64)
        cap = int(idx_cap)
        idx = np.empty(max(cap, 1), np.int32)
        m = C.c_size_t(0)
        self._check(self._lib.pie_scan(self._ctx, int(now), int(cutoff), _ptr(counts), _ptr(offsets), _ptr(idx), cap, C.byref(m)))
        return counts, offsets, idx[: m.value].copy() if cap > 4 * max(m.value, 1) else idx[: m.value]

    def read_results(self):
        """Host copies (counts, offsets, idx) of the last FINISHED scan (scan_finish / scan_device / scan_pipelined)."""
        U = self.n_users
        counts, offsets = np.empty(U, np.int32), np.empty(U + 1, np.int64)
        m = C.c_size_t(0)
        self._check(self._lib.pie_read_results(self._ctx, _ptr(counts), _ptr(offsets), None, 0, C.byref(m)))
        idx = np.empty(max(m.value, 1), np.int32)
        if m.value:
            self._check(self._lib.pie_read_results(self._ctx, None, None, _ptr(idx), m.value, C.byref(m)))
        return counts, offsets, idx[: m.value]

    def read_user_feed(self, user, cap=None):
        """Rows of one user's feed from the last finished scan (two small device reads)."""
        cap = self.n if cap is None else int(cap)
        out = np.empty(max(cap, 1), np.int32)
        k = C.c_size_t(0)
        self._check(self._lib.pie_read_user_feed(self._ctx, int(user), _ptr(out), cap, C.byref(k)))
        return out[: k.value].copy()

    def read_results_into(self, counts_ptr=None, offsets_ptr=None, idx_ptr=None, idx_cap=0):
        """pie_read_results into caller-owned host memory (raw addresses, e.g. of pinned buffers).  -> M."""
        m = C.c_size_t(0)
        self._check(self._lib.pie_read_results(self._ctx, counts_ptr, offsets_ptr, idx_ptr, int(idx_cap), C.byref(m)))
        return m.value

    def scan_device(self, now, cutoff):
        m = C.c_size_t(0)
        self._check(self._lib.pie_scan_device(self._ctx, int(now), int(cutoff), C.byref(m)))
        return m.value

    def scan_begin(self, now, cutoff):
        self._check(self._lib.pie_scan_begin(self._ctx, int(now), int(cutoff)))
        self._begun.append(False)

    def scan_finish(self):
        m = C.c_size_t(0)
        rc = self._lib.pie_scan_finish(self._ctx, C.byref(m))
        if self._begun and rc != -6:   # the library dropped the oldest scan (finished or failed); PIE_E_STATE: nothing was in flight
            self._begun.pop(0)
        self._check(rc)
        return m.value

    def scan_begin_packed(self, now, cutoff, dst_ptr, u_pad, idx_cap):
        """scan_begin whose scan also writes its result message (layout of pack_results_device) into dst_ptr."""
        self._check(self._lib.pie_scan_begin_packed(self._ctx, int(now), int(cutoff), dst_ptr, int(u_pad), int(idx_cap)))
        self._begun.append(True)

    def scan_begin_packed2(self, now, cutoff, dst_ptr, u_pad, idx_cap, counts_ptr=None):
        """scan_begin_packed with a second destination for counts[U]; both may be mapped pinned host memory (host_alloc)."""
        self._check(self._lib.pie_scan_begin_packed2(self._ctx, int(now), int(cutoff), dst_ptr, int(u_pad), int(idx_cap), counts_ptr))
        self._begun.append(True)

    def host_alloc(self, n_words):
        """Mapped pinned host memory of n_words int32.  -> (numpy view of the host side, device address, host address)."""
        h, d = _P(), _P()
        self._check(self._lib.pie_host_alloc(self._ctx, int(n_words) * 4, C.byref(h), C.byref(d)))
        arr = np.ctypeslib.as_array(C.cast(h.value, C.POINTER(C.c_int32)), shape=(int(n_words),))
        return arr, d.value, h.value

    def host_free(self, host_addr):
        self._check(self._lib.pie_host_free(self._ctx, host_addr))

    def set_scan_form(self, form):
        """Pin the table-pass form (pie_stats.k1_variant codes); form < 0: adaptive."""
        self._check(self._lib.pie_set_scan_form(self._ctx, int(form)))

    def set_ordered_run(self, mode):
        """0: never (frees the run), 1: adaptive (default), 2: always (pie_set_ordered_run)."""
        self._check(self._lib.pie_set_ordered_run(self._ctx, int(mode)))

    def table_info(self):
        ti = PieTableInfo()
        ti.struct_size = C.sizeof(PieTableInfo)
        self._check(self._lib.pie_table_info_get(self._ctx, C.byref(ti)))
        return {k: getattr(ti, k) for k, _ in PieTableInfo._fields_}

    def in_flight_packed(self):
        """True when the oldest scan in flight was begun with scan_begin_packed."""
        return bool(self._begun) and self._begun[0]

    def scan_finish_packed(self):
        """-> (M, ready): ready = the message was complete in device memory on return (no stream ordering needed);
        otherwise a pack kernel was enqueued on the context's stream."""
        m, ready = C.c_size_t(0), C.c_int(0)
        rc = self._lib.pie_scan_finish_packed(self._ctx, C.byref(m), C.byref(ready))
        if self._begun and rc != -6:
            self._begun.pop(0)
        self._check(rc)
        return int(m.value), bool(ready.value)

    def scan_pipelined(self, k, now, cutoff):
        """k scans of the same query with two in flight: the table pass of scan i+1 overlaps the tail (scatter,
        per-bucket order) of scan i.  -> M of the last scan; results on the device as after scan_device()."""
        m = 0
        if k <= 0:
            return m
        self.scan_begin(now, cutoff)
        for i in range(k):
            if i + 1 < k:
                self.scan_begin(now, cutoff)
            m = self.scan_finish()
        return m

    # ---- batched scan: Q queries (now, cutoff, mask), one table pass
    _q_cache = None

    @staticmethod
    def _queries(queries):
        # the same list of queries, batch after batch, is the common case (a pipelined loop): marshal it once
        hit = PieScan._q_cache
        if hit is not None and hit[2] is queries and len(queries) == len(hit[0]):
            return hit[1]            # the very same list object again (a caller that changes it in place passes a new list)
        key = tuple(queries)
        if hit is not None and hit[0] == key:
            return hit[1]
        arr = (PieQuery * len(queries))()
        for k, (now, cutoff, mask) in enumerate(queries):
            arr[k].now, arr[k].cutoff, arr[k].mask = int(now), int(cutoff), int(mask) & (2 ** 64 - 1)
        PieScan._q_cache = (key, arr, queries)
        return arr

    def scan_batch_begin(self, queries):
        """queries: sequence of (now, cutoff, mask), at most PIE_BATCH_MAX.  Up to three batches may be in flight."""
        arr = self._queries(queries)
        self._check(self._lib.pie_scan_batch_begin(self._ctx, arr, len(queries)))
        self._batches = getattr(self, "_batches", [])
        self._batches.append(len(queries))

    def scan_batch_begin_packed(self, queries, msg_ptr, msg_stride, u_pad, idx_cap, counts_ptr=None, counts_stride=0):
        arr = self._queries(queries)
        self._check(self._lib.pie_scan_batch_begin_packed(self._ctx, arr, len(queries), msg_ptr, int(msg_stride), int(u_pad),
                                                          int(idx_cap), counts_ptr, int(counts_stride)))
        self._batches = getattr(self, "_batches", [])
        self._batches.append(len(queries))

    def scan_batch_begin_union(self, queries, msg_ptr, u_pad, cap):
        """A batch whose own kernels write the UNION exchange message [uoff[0..u_pad] | Mu | rows[cap) | mask_lo[cap) | mask_hi[cap)
        if more than 32 queries] into msg_ptr (device-visible int32 memory)."""
        arr = self._queries(queries)
        self._check(self._lib.pie_scan_batch_begin_union(self._ctx, arr, len(queries), msg_ptr, int(u_pad), int(cap)))
        self._batches = getattr(self, "_batches", [])
        self._batches.append(len(queries))

    def batch_read_union(self):
        """The primary result of the last finished batch: (uoff[U+1] int64, rows[Mu] int32, masks[Mu] uint64);
        Feed(q, u) = rows[uoff[u]:uoff[u+1]][(masks[uoff[u]:uoff[u+1]] >> q) & 1 == 1].  None when the batch has no union
        (queries fell back to the general path, or it ran on the ordered run)."""
        a, b, lo, hi, mu = _P(), _P(), _P(), _P(), C.c_size_t(0)
        self._check(self._lib.pie_batch_union_device_ptrs(self._ctx, C.byref(a), C.byref(b), C.byref(lo), C.byref(hi), C.byref(mu)))
        if not a.value:
            return None
        uoff = np.empty(self.n_users + 1, np.int64)
        rows, masks = np.empty(max(mu.value, 1), np.int32), np.empty(max(mu.value, 1), np.uint64)
        self._check(self._lib.pie_batch_read_union(self._ctx, _ptr(uoff), _ptr(rows), _ptr(masks), mu.value, C.byref(mu)))
        return uoff, rows[: mu.value], masks[: mu.value]

    def batch_union_device_ptrs(self):
        """-> (uoff, rows, mask_lo, mask_hi device addresses or None, Mu)"""
        a, b, lo, hi, mu = _P(), _P(), _P(), _P(), C.c_size_t(0)
        self._check(self._lib.pie_batch_union_device_ptrs(self._ctx, C.byref(a), C.byref(b), C.byref(lo), C.byref(hi), C.byref(mu)))
        return a.value, b.value, lo.value, hi.value, int(mu.value)

    def scan_batch_finish(self, packed=False, want_m=True):
        """-> list of M per query of the oldest batch in flight (packed: (list, ready)).  want_m=False: the list is not built
        (a step loop that only moves messages: 64 Python ints per step are a microsecond it does not have)."""
        nq = self._batches[0] if getattr(self, "_batches", None) else PIE_BATCH_MAX
        m = self._m_buf
        ready = self._ready_buf
        rc = self._lib.pie_scan_batch_finish_packed(self._ctx, m, self._ready_ref)
        if getattr(self, "_batches", None) and rc != -6:
            self._batches.pop(0)
        if rc:
            self._check(rc)
        ms = m[:nq] if want_m else None
        return (ms, bool(ready.value)) if packed else ms

    def batch_pack_union_device(self, dst_ptr, u_pad, cap):
        """Union message of the last finished batch into device memory (pie_batch_pack_union_device); enqueued, not waited for."""
        self._check(self._lib.pie_batch_pack_union_device(self._ctx, dst_ptr, int(u_pad), int(cap)))

    def batch_read_results(self, qi):
        """Host copies (counts, offsets, idx) of query qi of the last finished batch."""
        U = self.n_users
        counts, offsets = np.empty(U, np.int32), np.empty(U + 1, np.int64)
        m = C.c_size_t(0)
        self._check(self._lib.pie_batch_read_results(self._ctx, int(qi), _ptr(counts), _ptr(offsets), None, 0, C.byref(m)))
        idx = np.empty(max(m.value, 1), np.int32)
        if m.value:
            self._check(self._lib.pie_batch_read_results(self._ctx, int(qi), None, None, _ptr(idx), m.value, C.byref(m)))
        return counts, offsets, idx[: m.value]

    def scan_batch(self, queries):
        """-> [(counts, offsets, idx)] per query; bit for bit what scan() gives for each (now, cutoff) under its mask."""
        self.scan_batch_begin(queries)
        self.scan_batch_finish()
        return [self.batch_read_results(k) for k in range(len(queries))]

    def batch_read_user_feed(self, qi, user, cap=None):
        """Rows of one user's feed of query qi of the last finished batch."""
        cap = self.n if cap is None else int(cap)
        out = np.empty(max(cap, 1), np.int32)
        k = C.c_size_t(0)
        self._check(self._lib.pie_batch_read_user_feed(self._ctx, int(qi), int(user), _ptr(out), cap, C.byref(k)))
        return out[: k.value].copy()

    def batch_fetch_requests(self, qis, users, cap_rows=None):
        """Feeds of many (query, user) requests of the last finished batch in one call.
        -> (off[n + 1] int64, idx, start, end, disc): request i's rows are idx[off[i]:off[i + 1]] (feed order) with their columns."""
        qis, users = _col(qis, np.int32), _col(users, np.int32)
        n = qis.shape[0]
        cap = int(cap_rows) if cap_rows is not None else 64 * max(n, 1)
        off = np.empty(n + 1, np.int64)
        idx, disc = np.empty(max(cap, 1), np.int32), np.empty(max(cap, 1), np.int32)
        start, end = np.empty(max(cap, 1), np.int64), np.empty(max(cap, 1), np.int64)
        total = C.c_size_t(0)
        self._check(self._lib.pie_batch_fetch_requests(self._ctx, _ptr(qis), _ptr(users), n, cap, _ptr(off), _ptr(idx), _ptr(start), _ptr(end),
                                                       _ptr(disc), C.byref(total)))
        t = total.value
        return off, idx[:t], start[:t], end[:t], disc[:t]

    def batch_result_device_ptrs(self, qi):
        a, b, c = _P(), _P(), _P()
        self._check(self._lib.pie_batch_result_device_ptrs(self._ctx, int(qi), C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def set_batch_lanes(self, n_lanes):
        """Lanes of the batched scan (independent streams, three batches in flight on each): 1..4, 0 = by table size."""
        self._check(self._lib.pie_set_batch_lanes(self._ctx, int(n_lanes)))

    def batch_lanes(self):
        return int(self._lib.pie_batch_lanes(self._ctx))

    def batch_room(self):
        """batches scan_batch_begin would take right now (pie_batch_room)"""
        return int(self._lib.pie_batch_room(self._ctx))

    def scan_batch_pipelined(self, k, queries, depth=3):
        """k batches of the same queries with up to `depth` (<= 3) in flight PER LANE: the next launch is queued before the host
        waits for a summary.  -> list of M of the last batch."""
        ms, begun, done = [], 0, 0
        depth = depth * self.batch_lanes()
        room = self._lib.pie_batch_room
        while done < k:
            while begun < k and begun - done < depth and room(self._ctx) > 0:
                self.scan_batch_begin(queries)
                begun += 1
                if begun == k:   # the burst ends here: the lanes' last tails go out together, not one by one as they are finished
                    self._check(self._lib.pie_scan_batch_flush(self._ctx))
            ms = self.scan_batch_finish()
            done += 1
        return ms

    def scan_batch_flush(self):
        """No further begin is coming for now: queue the waiting tails of all lanes at once (pie_scan_batch_flush)."""
        self._check(self._lib.pie_scan_batch_flush(self._ctx))

    def result_device_ptrs(self):
        a, b, c = _P(), _P(), _P()
        self._check(self._lib.pie_result_device_ptrs(self._ctx, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def copy_results_device(self, counts_ptr=None, offsets_ptr=None, idx_ptr=None, idx_cap=0):
        """D2D copy into caller-owned device buffers (raw pointers, e.g. tensor.data_ptr()), on the ctx stream."""
        self._check(self._lib.pie_copy_results_device(self._ctx, counts_ptr, offsets_ptr, idx_ptr, int(idx_cap)))

    def pack_results_device(self, dst_ptr, u_pad, idx_cap):
        """[off[0..u_pad] | M | idx[:min(M, cap)]] as int32 (u_pad+2+cap words) into caller-owned device memory."""
        self._check(self._lib.pie_pack_results_device(self._ctx, dst_ptr, int(u_pad), int(idx_cap)))

    def fetch_rows(self, idx):
        idx = _col(idx, np.int32)
        m = idx.shape[0]
        s, e = np.empty(m, np.int64), np.empty(m, np.int64)
        u, d = np.empty(m, np.int32), np.empty(m, np.int32)
        self._check(self._lib.pie_fetch_rows(self._ctx, _ptr(idx), m, _ptr(s), _ptr(e), _ptr(u), _ptr(d)))
        return s, e, u, d

    def expired_queue(self, prev_now, now, fetch=True):
        """Ascending rows with prev_now < end <= now.  fetch=False: leave the queue on the device, return its length."""
        q = C.c_size_t(0)
        if not fetch:
            self._check(self._lib.pie_expired_queue(self._ctx, int(prev_now), int(now), None, 0, C.byref(q)))
            return q.value
        out = np.empty(max(self.n, 1), np.int32)
        self._check(self._lib.pie_expired_queue(self._ctx, int(prev_now), int(now), _ptr(out), self.n, C.byref(q)))
        return out[: q.value].copy()

    def archive_queue(self, now, window_ms=43200000, fetch=True):
        """Rows of every group (user) whose earliest start is at least window_ms old, groups in first-appearance order.
        fetch=False: leave the queue on the device, return its length."""
        q = C.c_size_t(0)
        if not fetch:
            self._check(self._lib.pie_archive_queue(self._ctx, int(now), int(window_ms), None, 0, C.byref(q)))
            return q.value
        out = np.empty(max(self.n, 1), np.int32)
        self._check(self._lib.pie_archive_queue(self._ctx, int(now), int(window_ms), _ptr(out), self.n, C.byref(q)))
        return out[: q.value].copy()

    def archive_stats(self):
        """-> (device ms summed over the profiled archive chains, their number, algorithmic bytes of the last one)"""
        ms, calls, alg = C.c_double(0), C.c_uint32(0), C.c_uint64(0)
        self._check(self._lib.pie_archive_stats(self._ctx, C.byref(ms), C.byref(calls), C.byref(alg)))
        return ms.value, calls.value, alg.value

    # ---- sharding on the device
    def shard_table(self, rank, world):
        """Keep only the rows of the users that hash to `rank` of `world`, users re-numbered densely.  -> (n_rows, n_users)"""
        n, u = C.c_size_t(0), C.c_int32(0)
        self._check(self._lib.pie_shard_table(self._ctx, int(rank), int(world), C.byref(n), C.byref(u)))
        self.n, self.n_users = int(n.value), int(u.value)
        return self.n, self.n_users

    def shard_maps(self, n_users_real=None):
        """-> (rows_global[n] int32, users_global[k] int32): local row -> global row, local user -> global user."""
        rows = np.empty(max(self.n, 1), np.int32)
        users = np.full(max(self.n_users, 1), -1, np.int32)
        self._check(self._lib.pie_shard_maps(self._ctx, _ptr(rows), _ptr(users)))
        return rows[: self.n], users

    # ---- measurement / plumbing
    def set_stream(self, hip_stream):
        self._check(self._lib.pie_ctx_set_stream(self._ctx, hip_stream))

    def aux_stream(self):
        """hipStream_t (as int) on which scan tails and result copies / packs run."""
        p = _P()
        self._check(self._lib.pie_ctx_aux_stream(self._ctx, C.byref(p)))
        return p.value

    def set_profiling(self, every=1):
        """0/False: off; n: every n-th scan carries timing events."""
        self._check(self._lib.pie_set_profiling(self._ctx, int(every)))

    def stats(self):
        st = PieStats()
        st.struct_size = C.sizeof(PieStats)
        self._check(self._lib.pie_stats_get(self._ctx, C.byref(st)))
        return {k: getattr(st, k) for k, _ in PieStats._fields_}

    def stats_reset(self):
        self._check(self._lib.pie_stats_reset(self._ctx))

    def synchronize(self):
        self._check(self._lib.pie_synchronize(self._ctx))


class PieComm:
    """The sharded table behind the C ABI: one scan context per GPU + an RCCL communicator (pie_comm_*).
    PieComm(device_ids) = one process drives all GPUs; PieComm.for_rank(id, rank, world, device) = one process per GPU."""

    def __init__(self, device_ids=None, _handle=None):
        self._lib = load_library()
        self._c = _P()
        if _handle is not None:
            self._c = _handle
        else:
            ids = (C.c_int32 * len(device_ids))(*[int(d) for d in device_ids])
            rc = self._lib.pie_comm_create(ids, len(device_ids), C.byref(self._c))
            if rc != 0:
                text = self._lib.pie_comm_last_error(None).decode()
                self._c = None
                raise PieError(rc, text)
        self.world = int(self._lib.pie_comm_world(self._c))

    @staticmethod
    def unique_id():
        lib = load_library()
        buf = C.create_string_buffer(128)
        rc = lib.pie_comm_unique_id(buf)
        if rc != 0:
            raise PieError(rc, lib.pie_comm_last_error(None).decode())
        return buf.raw

    @classmethod
    def for_rank(cls, unique_id, rank, world, device):
        lib = load_library()
        h = _P()
        rc = lib.pie_comm_create_rank(C.c_char_p(unique_id), int(rank), int(world), int(device), C.byref(h))
        if rc != 0:
            raise PieError(rc, lib.pie_comm_last_error(None).decode())
        return cls(_handle=h)

    def close(self):
        if getattr(self, "_c", None):
            self._lib.pie_comm_destroy(self._c)
            self._c = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise PieError(rc, self._lib.pie_comm_last_error(self._c).decode())

    def ctx(self, rank):
        """The shard's scan context as a PieScan (owned by the communicator: do not close it)."""
        h = self._lib.pie_comm_ctx(self._c, int(rank))
        if not h:
            raise PieError(-1, "rank %d is not local to this communicator" % rank)
        p = PieScan.__new__(PieScan)
        p._lib, p._ctx, p._begun = self._lib, _P(h), []
        p._m_buf, p._ready_buf = (C.c_size_t * PIE_BATCH_MAX)(), C.c_int(0)
        p._ready_ref = C.byref(p._ready_buf)
        st = PieStats()
        st.struct_size = C.sizeof(PieStats)
        p._check(self._lib.pie_stats_get(p._ctx, C.byref(st)))
        p.n, p.n_users = int(st.rows), int(st.users)
        p.close = lambda: None
        return p

    def gen_synthetic_sharded(self, seed, n_total, n_users, n_disc, flags=0):
        self._check(self._lib.pie_comm_gen_synthetic_sharded(self._c, seed, n_total, n_users, n_disc, flags))

    def reserve(self, n_q, u_pad, idx_cap):
        self._check(self._lib.pie_comm_reserve(self._c, int(n_q), int(u_pad), int(idx_cap)))

    def scan_batch_gather(self, queries, u_pad=0):
        """-> M[local rank][query]"""
        arr = PieScan._queries(queries)
        nl = int(self._lib.pie_comm_local_ranks(self._c))
        m = (C.c_size_t * (nl * len(queries)))()
        self._check(self._lib.pie_comm_scan_batch_gather(self._c, arr, len(queries), int(u_pad), m))
        return [[int(m[k * len(queries) + q]) for q in range(len(queries))] for k in range(nl)]

    # ---- the pipelined union exchange (pie_comm_step_*)
    def needed_cap(self):
        return int(self._lib.pie_comm_needed_cap(self._c))

    def step_reserve(self, n_q, u_pad=0, union_cap=1024):
        self._check(self._lib.pie_comm_step_reserve(self._c, int(n_q), int(u_pad), int(union_cap)))

    def step_begin(self, queries):
        arr = PieScan._queries(queries)
        self._check(self._lib.pie_comm_step_begin(self._c, arr, len(queries)))
        self._step_nq = getattr(self, "_step_nq", [])
        self._step_nq.append(len(queries))

    def step_finish(self):
        """-> M[local rank][query] of the oldest begun step; its exchange is queued, not waited for."""
        nq = self._step_nq.pop(0)
        nl = int(self._lib.pie_comm_local_ranks(self._c))
        m = (C.c_size_t * (nl * nq))()
        self._check(self._lib.pie_comm_step_finish(self._c, m))
        return [[int(m[k * nq + q]) for q in range(nq)] for k in range(nl)]

    def step_collect(self):
        """Wait for the oldest queued exchange.  -> its step number; PieError(PIE_E_CAPACITY) on every rank alike when a union
        outgrew the reserved capacity (needed_cap() says what to reserve)."""
        step = C.c_int64(-1)
        self._check(self._lib.pie_comm_step_collect(self._c, C.byref(step)))
        return int(step.value)

    def step_read_gathered(self, at_rank, src_rank, step):
        """-> (uoff[u_pad + 1] int32, rows[Mu] int32, masks[Mu] uint64) of src_rank's union message of `step` as at_rank holds it."""
        base, rs, up, cap = _P(), C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        self._check(self._lib.pie_comm_step_gathered_ptr(self._c, int(at_rank), int(step), C.byref(base), C.byref(rs), C.byref(up), C.byref(cap)))
        uoff = np.empty(up.value + 1, np.int32)
        mu = C.c_size_t(0)
        self._check(self._lib.pie_comm_step_read_gathered(self._c, int(at_rank), int(src_rank), int(step), _ptr(uoff), None, None, 0, C.byref(mu)))
        rows, masks = np.empty(max(mu.value, 1), np.int32), np.empty(max(mu.value, 1), np.uint64)
        self._check(self._lib.pie_comm_step_read_gathered(self._c, int(at_rank), int(src_rank), int(step), None, _ptr(rows), _ptr(masks), mu.value, C.byref(mu)))
        return uoff, rows[: mu.value], masks[: mu.value]

    def read_gathered(self, at_rank, src_rank, qi):
        """-> (offsets[u_pad + 1] int32, idx[M] int32) of (src_rank, query qi) as rank at_rank holds it."""
        base, rs, qs, up = _P(), C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        self._check(self._lib.pie_comm_gathered_device_ptr(self._c, int(at_rank), C.byref(base), C.byref(rs), C.byref(qs), C.byref(up)))
        off = np.empty(up.value + 1, np.int32)
        m = C.c_size_t(0)
        self._check(self._lib.pie_comm_read_gathered(self._c, int(at_rank), int(src_rank), int(qi), _ptr(off), None, 0, C.byref(m)))
        idx = np.empty(max(m.value, 1), np.int32)
        self._check(self._lib.pie_comm_read_gathered(self._c, int(at_rank), int(src_rank), int(qi), None, _ptr(idx), m.value, C.byref(m)))
        return off, idx[: m.value]


def shard_of(user, n_shards):
    return load_library().pie_shard_of(int(user), int(n_shards))


def tz_table(zone, from_ms=0, to_ms=4102444800000):
    """Transition table of an IANA zone for retention_purge(tz_table=...): (transitions_utc_ms[n], offsets_ms[n + 1]), from
    Python's zoneinfo, probed day by day and bisected to the millisecond (the Node host builds the same table from the JS
    engine: host/tzTable.js)."""
    import datetime
    import zoneinfo
    z = zoneinfo.ZoneInfo(zone)
    epoch = datetime.datetime(1970, 1, 1, tzinfo=datetime.timezone.utc)
    ms1 = datetime.timedelta(milliseconds=1)

    def off(ms):
        return int((epoch + datetime.timedelta(milliseconds=int(ms))).astimezone(z).utcoffset() / ms1)

    day = 86400000
    trans, offs = [], [off(from_ms)]
    prev_t, prev_o, t = from_ms, offs[0], from_ms + day
    while prev_t < to_ms:
        at = min(t, to_ms)
        o = off(at)
        if o != prev_o:
            lo, hi = prev_t, at
            while hi - lo > 1:
                mid = (lo + hi) // 2
                if off(mid) == prev_o:
                    lo = mid
                else:
                    hi = mid
            trans.append(hi)
            offs.append(o)
            prev_o = o
        prev_t = at
        t += day
    return np.array(trans, np.int64), np.array(offs, np.int64)
