"""TEST INFRASTRUCTURE — ctypes wrapper of oracle/libpie_oracle.so (the CPU restatement, oracle/pie_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product package
(sph-pie_amd/) never does."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libpie_oracle.so")

T0_MS = 1700000000000
SPAN_MS = 10368000000
TTL_MS = 43200000
GEN_INTERVAL = 1
GEN_CLUSTERED = 2
GEN_TIME_ORDERED = 4
INT64_MIN = -(2 ** 63)

_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "pie_oracle.c")
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", HERE, "-B", "libpie_oracle.so"], stdout=subprocess.DEVNULL)
        l = C.CDLL(LIB)
        P = C.c_void_p
        l.pie_oracle_gen.restype = None
        l.pie_oracle_gen.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_uint32, P, P, P, P]
        l.pie_oracle_gen_cdf.restype = None
        l.pie_oracle_gen_cdf.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_uint32, P, P, P, P, P]
        l.pie_oracle_selected.restype = C.c_int
        l.pie_oracle_selected.argtypes = [C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_uint64]
        l.pie_oracle_scan.restype = C.c_int
        l.pie_oracle_scan.argtypes = [P, P, P, P, C.c_size_t, C.c_int32, C.c_int64, C.c_int64, C.c_uint64, P, P, P,
                                      C.c_size_t, C.POINTER(C.c_size_t)]
        l.pie_oracle_scan_mt.restype = C.c_int
        l.pie_oracle_scan_mt.argtypes = [P, P, P, P, C.c_size_t, C.c_int32, C.c_int64, C.c_int64, C.c_uint64, P, P, P,
                                         C.c_size_t, C.POINTER(C.c_size_t), C.c_int]
        l.pie_oracle_gen_mt.restype = None
        l.pie_oracle_gen_mt.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_uint32, P, P, P, P, C.c_int]
        l.pie_oracle_expired_queue.restype = C.c_int
        l.pie_oracle_expired_queue.argtypes = [P, C.c_size_t, C.c_int64, C.c_int64, P, C.c_size_t, C.POINTER(C.c_size_t)]
        l.pie_oracle_archive_queue.restype = C.c_int
        l.pie_oracle_archive_queue.argtypes = [P, P, P, C.c_size_t, C.c_int32, C.c_int64, C.c_int64, P, C.c_size_t, C.POINTER(C.c_size_t)]
        l.pie_oracle_add_months.restype = C.c_int64
        l.pie_oracle_add_months.argtypes = [C.c_int64, C.c_int32, C.c_int64, C.POINTER(C.c_int)]
        l.pie_oracle_add_months_tz.restype = C.c_int64
        l.pie_oracle_add_months_tz.argtypes = [C.c_int64, C.c_int32, P, P, C.c_int32, C.POINTER(C.c_int)]
        l.pie_oracle_retention_queue_tz.restype = C.c_int
        l.pie_oracle_retention_queue_tz.argtypes = [P, P, C.c_size_t, C.c_int64, C.c_int32, P, P, C.c_int32, P, C.c_size_t, C.POINTER(C.c_size_t)]
        l.pie_oracle_retention_queue.restype = C.c_int
        l.pie_oracle_retention_queue.argtypes = [P, P, C.c_size_t, C.c_int64, C.c_int32, C.c_int64, P, C.c_size_t, C.POINTER(C.c_size_t)]
        l.pie_oracle_splitmix64.restype = C.c_uint64
        l.pie_oracle_splitmix64.argtypes = [C.c_uint64]
        l.pie_oracle_shard_of.restype = C.c_int32
        l.pie_oracle_shard_of.argtypes = [C.c_int32, C.c_int32]
        _lib = l
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def gen(seed, n_total, row0, n, n_users, n_disc, flags=0):
    s, e = np.empty(n, np.int64), np.empty(n, np.int64)
    u, d = np.empty(n, np.int32), np.empty(n, np.int32)
    lib().pie_oracle_gen(seed, n_total, row0, n, n_users, n_disc, flags, _p(s), _p(e), _p(u), _p(d))
    return s, e, u, d


def zipf_cdf(n_users, exponent=1.1):
    """uint64 thresholds floor(CDF_k * 2^64) of Zipf(exponent) over n_users ranks; the last one is 2^64 - 1."""
    w = 1.0 / np.arange(1, n_users + 1, dtype=np.float64) ** exponent
    c = np.cumsum(w) / w.sum()
    thr = np.minimum(np.floor(c * 2.0 ** 64), 2.0 ** 64 - 2048).astype(np.uint64)
    thr[-1] = np.uint64(2 ** 64 - 1)
    return np.maximum.accumulate(thr)


def gen_cdf(seed, n_total, row0, n, n_users, n_disc, flags, cdf):
    s, e = np.empty(n, np.int64), np.empty(n, np.int64)
    u, d = np.empty(n, np.int32), np.empty(n, np.int32)
    cdf = np.ascontiguousarray(cdf, np.uint64)
    assert cdf.shape[0] == n_users
    lib().pie_oracle_gen_cdf(seed, n_total, row0, n, n_users, n_disc, flags, _p(cdf), _p(s), _p(e), _p(u), _p(d))
    return s, e, u, d


def scan(start, end, user, disc, n_users, now, cutoff, mask):
    start, end = np.ascontiguousarray(start, np.int64), np.ascontiguousarray(end, np.int64)
    user, disc = np.ascontiguousarray(user, np.int32), np.ascontiguousarray(disc, np.int32)
    n = start.shape[0]
    counts, offsets = np.empty(n_users, np.int32), np.empty(n_users + 1, np.int64)
    idx = np.empty(max(n, 1), np.int32)
    m = C.c_size_t(0)
    rc = lib().pie_oracle_scan(_p(start), _p(end), _p(user), _p(disc), n, n_users, int(now), int(cutoff),
                               int(mask) & (2 ** 64 - 1), _p(counts), _p(offsets), _p(idx), n, C.byref(m))
    if rc != 0:
        raise RuntimeError("pie_oracle_scan rc=%d" % rc)
    return counts, offsets, idx[: m.value].copy()


def gen_mt(seed, n_total, row0, n, n_users, n_disc, flags=0, threads=1):
    """gen() on `threads` host threads (row slices); same corpus."""
    s, e = np.empty(n, np.int64), np.empty(n, np.int64)
    u, d = np.empty(n, np.int32), np.empty(n, np.int32)
    lib().pie_oracle_gen_mt(seed, n_total, row0, n, n_users, n_disc, flags, _p(s), _p(e), _p(u), _p(d), int(threads))
    return s, e, u, d


def scan_mt(start, end, user, disc, n_users, now, cutoff, mask, threads, out=None):
    """scan() on `threads` host threads (bench.py's cpu_baseline B2); identical output.  `out` = (counts, offsets, idx)
    buffers to reuse between repetitions (idx sized n)."""
    n = start.shape[0]
    if out is None:
        out = (np.empty(n_users, np.int32), np.empty(n_users + 1, np.int64), np.empty(max(n, 1), np.int32))
    counts, offsets, idx = out
    m = C.c_size_t(0)
    rc = lib().pie_oracle_scan_mt(_p(start), _p(end), _p(user), _p(disc), n, n_users, int(now), int(cutoff),
                                  int(mask) & (2 ** 64 - 1), _p(counts), _p(offsets), _p(idx), n, C.byref(m), int(threads))
    if rc != 0:
        raise RuntimeError("pie_oracle_scan_mt rc=%d" % rc)
    return counts, offsets, idx[: m.value]


def expired_queue(end, prev_now, now):
    end = np.ascontiguousarray(end, np.int64)
    n = end.shape[0]
    q = np.empty(max(n, 1), np.int32)
    k = C.c_size_t(0)
    rc = lib().pie_oracle_expired_queue(_p(end), n, int(prev_now), int(now), _p(q), n, C.byref(k))
    if rc != 0:
        raise RuntimeError("pie_oracle_expired_queue rc=%d" % rc)
    return q[: k.value].copy()


def archive_queue(start, end, user, n_users, now, window_ms):
    start, end = np.ascontiguousarray(start, np.int64), np.ascontiguousarray(end, np.int64)
    user = np.ascontiguousarray(user, np.int32)
    n = start.shape[0]
    q = np.empty(max(n, 1), np.int32)
    k = C.c_size_t(0)
    rc = lib().pie_oracle_archive_queue(_p(start), _p(end), _p(user), n, n_users, int(now), int(window_ms), _p(q), n, C.byref(k))
    if rc != 0:
        raise RuntimeError("pie_oracle_archive_queue rc=%d" % rc)
    return q[: k.value].copy()


def archive_queue_numpy(start, end, user, n_users, now, window_ms):
    """Independent restatement (numpy) of the same chain, fast enough for 10^6 rows."""
    start, end, user = np.asarray(start, np.int64), np.asarray(end, np.int64), np.asarray(user, np.int64)
    rows = np.nonzero(end != INT64_MIN)[0]
    if rows.size == 0:
        return np.zeros(0, np.int32)
    g = user[rows]
    earliest = np.full(n_users, np.iinfo(np.int64).max, np.int64)
    np.minimum.at(earliest, g, start[rows])
    first = np.full(n_users, np.iinfo(np.int64).max, np.int64)
    np.minimum.at(first, g, rows)
    qual = np.array([(int(now) - int(e)) >= int(window_ms) if e != np.iinfo(np.int64).max else False for e in earliest])
    keep = rows[qual[g]]
    order = np.lexsort((keep, first[user[keep]]))
    return keep[order].astype(np.int32)


def add_months(ts, months, tz_offset_ms=0):
    """-> int, or None when the JS result would be NaN"""
    nan = C.c_int(0)
    v = lib().pie_oracle_add_months(int(ts), int(months), int(tz_offset_ms), C.byref(nan))
    return None if nan.value else v


def add_months_tz(ts, months, transitions, offsets):
    """setMonth on a local Date under the zone given as a transition table (see pie_oracle.h).  -> int or None (NaN)"""
    T, off = np.ascontiguousarray(transitions, np.int64), np.ascontiguousarray(offsets, np.int64)
    nan = C.c_int(0)
    v = lib().pie_oracle_add_months_tz(int(ts), int(months), _p(T), _p(off), int(T.size), C.byref(nan))
    return None if nan.value else v


def retention_queue_tz(start, end, now, months, transitions, offsets):
    start, end = np.ascontiguousarray(start, np.int64), np.ascontiguousarray(end, np.int64)
    T, off = np.ascontiguousarray(transitions, np.int64), np.ascontiguousarray(offsets, np.int64)
    n = start.shape[0]
    q = np.empty(max(n, 1), np.int32)
    k = C.c_size_t(0)
    rc = lib().pie_oracle_retention_queue_tz(_p(start), _p(end), n, int(now), int(months), _p(T), _p(off), int(T.size), _p(q), n, C.byref(k))
    if rc != 0:
        raise RuntimeError("pie_oracle_retention_queue_tz rc=%d" % rc)
    return q[: k.value].copy()


def tz_table(zone, from_ms=0, to_ms=4102444800000):
    """The transition table of `zone` from Python's zoneinfo (a different engine and data set from the JS one that produced the
    golden tables): (transitions[n], offsets[n + 1]) in ms, probed day by day and bisected to the millisecond."""
    import datetime
    import zoneinfo
    z = zoneinfo.ZoneInfo(zone)
    epoch = datetime.datetime(1970, 1, 1, tzinfo=datetime.timezone.utc)

    def off(ms):
        d = (epoch + datetime.timedelta(milliseconds=int(ms))).astimezone(z)
        return int(d.utcoffset() / datetime.timedelta(milliseconds=1))

    day = 86400000
    T, O = [], [off(from_ms)]
    prev_t, prev_o = from_ms, O[0]
    t = from_ms + day
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
            T.append(hi)
            O.append(o)
            prev_o = o
        prev_t = at
        t += day
    return np.array(T, np.int64), np.array(O, np.int64)


def calendar_cutoff(now_ms, months_back=2, tz="UTC"):
    """Restatement of the window scalar, /root/reference/server/calendarFeed.js:33-38: local midnight of `now` in zone
    `tz`, minus months_back calendar months with JS Date rules — the month shift is applied to the LOCAL fields of the
    midnight instant (on a day whose midnight does not exist, a DST change at 00:00, that instant reads 01:00 and the
    hour is kept), day-of-month overflow rolls into the following month (Apr 30 - 2 -> "Feb 30" -> Mar 2).  A different
    engine (Python zoneinfo) from the one that produced the pin, tests/golden/cutoff_zones.json (JS Date, oracle/
    gen_cutoff_golden.js)."""
    import datetime
    import zoneinfo
    z = zoneinfo.ZoneInfo(tz)
    epoch = datetime.datetime(1970, 1, 1, tzinfo=datetime.timezone.utc)
    at = lambda ms: (epoch + datetime.timedelta(milliseconds=int(ms))).astimezone(z)
    to_ms = lambda d: (d - epoch) // datetime.timedelta(milliseconds=1)
    dt = at(now_ms)
    midnight = datetime.datetime(dt.year, dt.month, dt.day, tzinfo=z)           # setHours(0, 0, 0, 0)
    lt = at(to_ms(midnight))                                                    # what the Date now reads in local time
    mi = (lt.month - 1) - int(months_back)                                      # setMonth(getMonth() - monthsBack)
    y = lt.year + mi // 12
    mi %= 12
    day = datetime.date(y, mi + 1, 1) + datetime.timedelta(days=lt.day - 1)     # MakeDay: day overflow rolls over
    local = datetime.datetime(day.year, day.month, day.day, lt.hour, lt.minute, lt.second, tzinfo=z)
    return to_ms(local)


def retention_queue(start, end, now, months=2, tz_offset_ms=0):
    start, end = np.ascontiguousarray(start, np.int64), np.ascontiguousarray(end, np.int64)
    n = start.shape[0]
    q = np.empty(max(n, 1), np.int32)
    k = C.c_size_t(0)
    rc = lib().pie_oracle_retention_queue(_p(start), _p(end), n, int(now), int(months), int(tz_offset_ms), _p(q), n, C.byref(k))
    if rc != 0:
        raise RuntimeError("pie_oracle_retention_queue rc=%d" % rc)
    return q[: k.value].copy()


def selected(start, end, disc, now, cutoff, mask):
    return bool(lib().pie_oracle_selected(int(start), int(end), int(disc), int(now), int(cutoff), int(mask) & (2 ** 64 - 1)))


def shard_of(user, n_shards):
    return lib().pie_oracle_shard_of(int(user), int(n_shards))


def scan_numpy(start, end, user, disc, n_users, now, cutoff, mask):
    """Independent second restatement in numpy (lexsort), used to cross-check the C oracle itself."""
    start, end = np.asarray(start, np.int64), np.asarray(end, np.int64)
    user, disc = np.asarray(user, np.int32), np.asarray(disc, np.int32)
    d_ok = (disc >= 0) & (disc < 64)
    bit = np.zeros(disc.shape, bool)
    dd = np.where(d_ok, disc, 0).astype(np.uint64)
    bit[d_ok] = ((np.uint64(int(mask) & (2 ** 64 - 1)) >> dd[d_ok]) & np.uint64(1)).astype(bool)
    sel = (end > now) & (start >= cutoff) & bit
    rows = np.nonzero(sel)[0].astype(np.int64)
    order = np.lexsort((rows, start[rows], user[rows]))  # last key is primary
    idx = rows[order].astype(np.int32)
    counts = np.bincount(user[rows], minlength=n_users).astype(np.int32)
    offsets = np.zeros(n_users + 1, np.int64)
    np.cumsum(counts, out=offsets[1:])
    return counts, offsets, idx
