"""CPU: pin the oracle (oracle/pie_oracle.c) against outputs of the real reference module
server/sessionStore.js (tests/golden/sessionstore_g1_g4.json, made by oracle/gen_golden.js) and against the
hand-derived vectors (tests/golden/hand_derived_h1_h5.json)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

INT64_MIN = -(2 ** 63)
ALL = 2 ** 64 - 1


@pytest.fixture(scope="module")
def g():
    with open(os.path.join(GOLDEN, "sessionstore_g1_g4.json")) as f:
        return json.load(f)


def columns(g):
    users = g["users"]
    rows = g["sessions"]
    start = np.array([r["createdAt"] for r in rows], np.int64)
    end = np.array([r["expiresAt"] for r in rows], np.int64)
    user = np.array([users.index(r["user"]) for r in rows], np.int32)
    disc = np.zeros(len(rows), np.int32)
    return start, end, user, disc, len(users)


def test_schema_constants(g):
    assert g["ttl_ms"] == 43200000 and g["cookie_name"] == "mt_session"
    s, e, _, _, _ = columns(g)
    assert np.all(e - s == g["ttl_ms"])  # expiresAt = createdAt + SESSION_TTL_MS (sessionStore.js:15-16)
    assert g["G1_falsy_token_null"] is True


def test_g1_liveness(g, oracle):
    s, e, u, d, U = columns(g)
    assert len(g["G1"]) >= 15
    for case in g["G1"]:
        counts, offsets, idx = oracle.scan(s, e, u, d, U, case["now"], INT64_MIN, ALL)
        live = np.zeros(len(s), int)
        live[idx] = 1
        assert live.tolist() == case["live"], case["now"]
        for i in range(len(s)):
            assert oracle.selected(s[i], e[i], 0, case["now"], INT64_MIN, 1) == bool(case["live"][i])


def test_g2_purge(g, oracle):
    s, e, u, d, U = columns(g)
    for case in g["G2"]:
        _, _, idx = oracle.scan(s, e, u, d, U, case["now"], INT64_MIN, ALL)
        assert sorted(idx.tolist()) == case["survivors"]
        # the complement is exactly what purge deleted == expired queue since forever
        q = oracle.expired_queue(e, INT64_MIN, case["now"])
        assert sorted(set(range(len(s))) - set(q.tolist())) == case["survivors"]


def test_g3_user_match(g, oracle):
    s, e, u, d, U = columns(g)
    users = g["users"]
    for case in g["G3"]:
        counts, offsets, idx = oracle.scan(s, e, u, d, U, case["observe_now"], INT64_MIN, ALL)
        assert len(idx) == len(s)  # nothing expired at observe_now
        name = case["user"]
        if name in users:
            k = users.index(name)
            feed = idx[offsets[k]:offsets[k + 1]].tolist()
            assert sorted(set(range(len(s))) - set(feed)) == case["survivors"]
            assert counts[k] == len(feed) > 0
        else:  # unknown or falsy id: no-op (sessionStore.js:56-58)
            assert case["survivors"] == list(range(len(s)))


def test_g4_touch(g, oracle):
    s, e, u, d, U = columns(g)
    for t in g["G4"]:
        k = t["row"]
        if t["returned"] is None:
            assert not oracle.selected(s[k], e[k], 0, t["now"], INT64_MIN, 1)
            continue
        assert oracle.selected(s[k], e[k], 0, t["now"], INT64_MIN, 1)
        assert t["returned"]["expiresAt"] == t["now"] + g["ttl_ms"] == t["after"]["expiresAt"]
        assert t["after"]["createdAt"] == s[k]


def test_feed_order_is_start_then_row(g, oracle):
    s, e, u, d, U = columns(g)
    counts, offsets, idx = oracle.scan(s, e, u, d, U, int(s.min()), INT64_MIN, ALL)
    for k in range(U):
        feed = idx[offsets[k]:offsets[k + 1]]
        keys = [(int(s[i]), int(i)) for i in feed]
        assert keys == sorted(keys)
        assert np.all(u[feed] == k)
    # rows 72..74 were created in the same millisecond: two of them by users[0] -> tie resolved by row index
    f0 = idx[offsets[0]:offsets[1]].tolist()
    assert f0.index(72) < f0.index(74)


# ---------------------------------------------------------------- hand-derived vectors (no executable reference)

@pytest.fixture(scope="module")
def h():
    with open(os.path.join(GOLDEN, "hand_derived_h1_h5.json")) as f:
        return json.load(f)


def test_h_vectors(h, oracle):
    assert "hand-derived" in h["provenance"]
    for case in h["scan_cases"]:
        c = case["columns"]
        counts, offsets, idx = oracle.scan(c["start"], c["end"], c["user"], c["disc"], case["n_users"], case["now"],
                                           case["cutoff"], case["mask"])
        assert counts.tolist() == case["expect"]["counts"], case["name"]
        assert offsets.tolist() == case["expect"]["offsets"], case["name"]
        assert idx.tolist() == case["expect"]["idx"], case["name"]
        c2, o2, i2 = oracle.scan_numpy(c["start"], c["end"], c["user"], c["disc"], case["n_users"], case["now"],
                                       case["cutoff"], case["mask"])
        assert (c2.tolist(), o2.tolist(), i2.tolist()) == (counts.tolist(), offsets.tolist(), idx.tolist())


@pytest.mark.parametrize("n,U,D,flags", [(1000, 10, 3, 0), (5000, 7, 32, 1), (4096, 64, 5, 2), (20000, 3, 64, 3)])
def test_c_oracle_matches_numpy_restatement(oracle, n, U, D, flags):
    s, e, u, d = oracle.gen(0x5EED5EED, n, 0, n, U, D, flags)
    assert u.min() >= 0 and u.max() < U and d.min() >= 0 and d.max() < D
    if flags & 2:
        assert np.all(np.diff(u) >= 0)
    for now, cutoff, mask in [(oracle.T0_MS - 6 * 3600 * 1000, oracle.T0_MS - 61 * 86400 * 1000, 0x5555555555555555),
                              (INT64_MIN, INT64_MIN, ALL), (2 ** 62, INT64_MIN, ALL),
                              (oracle.T0_MS - 100 * 86400 * 1000, oracle.T0_MS - 61 * 86400 * 1000, 0xAAAAAAAAAAAAAAAA)]:
        a = oracle.scan(s, e, u, d, U, now, cutoff, mask)
        b = oracle.scan_numpy(s, e, u, d, U, now, cutoff, mask)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_generator_slices_agree(oracle):
    full = oracle.gen(0x5EED5EED, 3000, 0, 3000, 17, 5, 1)
    part = oracle.gen(0x5EED5EED, 3000, 1000, 500, 17, 5, 1)
    for a, b in zip(full, part):
        assert np.array_equal(a[1000:1500], b)
    # published splitmix64 known answers (seed 0): first outputs of the sequential generator
    l = oracle.lib()
    assert l.pie_oracle_splitmix64(0) == 0xE220A8397B1DCDAF
    assert l.pie_oracle_splitmix64(0x9E3779B97F4A7C15) == 0x6E789E6AA1B965F4


def test_oracle_under_sanitizers(tmp_path):
    """ASan + UBSan over the native CPU code (the only native code a sanitizer can cover here: GPU ASan is not
    available on the pool).  Sizes straddle the merge-sort run length and the realloc growth points."""
    import shutil
    import subprocess
    from conftest import REPO
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    odir = os.path.join(REPO, "oracle")
    exe = tmp_path / "selftest_asan"
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-Wall", "-Wextra", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-o", str(exe), os.path.join(odir, "selftest.c"),
                           os.path.join(odir, "pie_oracle.c")])
    res = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0 and "oracle selftest ok" in res.stdout, res.stdout


def test_archive_chain_hand_vector_and_twin(oracle):
    """sqlProvider.js:758-816 restated: hand-worked vector (groups in first-appearance order, rows in table order,
    threshold inclusive) + the C oracle against its independent numpy twin."""
    u = [2, 0, 2, 1, 0]
    s = [100, 50, 90, 300, 60]
    e = [10 ** 12] * 5
    # W = 10: earliest g2 = 90, g0 = 50, g1 = 300; at now = 105 g2 and g0 qualify (15, 55 >= 10), g1 does not
    assert oracle.archive_queue(s, e, u, 3, 105, 10).tolist() == [0, 2, 1, 4]
    assert oracle.archive_queue(s, e, u, 3, 99, 10).tolist() == [1, 4]        # 99 - 90 = 9 < 10: g2 waits
    assert oracle.archive_queue(s, e, u, 3, 100, 10).tolist() == [0, 2, 1, 4]  # == window: inclusive (:798 `>=`)
    assert oracle.archive_queue(s, e, u, 3, 310, 10).tolist() == [0, 2, 1, 4, 3]
    e2 = list(e)
    e2[2] = INT64_MIN                                                          # the row holding g2's minimum is gone
    assert oracle.archive_queue(s, e2, u, 3, 105, 10).tolist() == [1, 4]       # g2's earliest is now 100: 5 < 10
    rng = np.random.default_rng(5)
    for n, U in [(0, 3), (1, 1), (2000, 11), (20000, 700)]:
        s, e, u, d = oracle.gen(0x5EED5EED, max(n, 1), 0, n, U, 3, 1)
        if n:
            e[rng.random(n) < 0.1] = INT64_MIN
        for now in [oracle.T0_MS, oracle.T0_MS - 119 * 86400000, INT64_MIN, 2 ** 62]:
            assert np.array_equal(oracle.archive_queue(s, e, u, U, now, 43200000),
                                  oracle.archive_queue_numpy(s, e, u, U, now, 43200000))


def test_add_months_matches_js_date_vectors(oracle):
    """The libc-based add-months oracle against 927 results of the JS engine's Date (TZ=UTC): month-end overflow, leap
    years, negative timestamps, year wrap, the +-8.64e15 range rules of sqlProvider.js:999-1009."""
    g = json.load(open(os.path.join(GOLDEN, "addmonths_utc.json")))
    assert len(g["cases"]) > 900
    for ts, m, want in g["cases"]:
        assert oracle.add_months(ts, m, 0) == want, (ts, m)
    # hand-checked: Dec 31 + 2 months overflows February (2025: Mar 3; leap 2024: Mar 2), sqlProvider.js:1007
    assert oracle.add_months(1735603200000, 2) == 1740960000000       # 2024-12-31T00:00Z -> 2025-03-03T00:00Z
    assert oracle.add_months(1703980800000, 2) == 1709337600000       # 2023-12-31T00:00Z -> 2024-03-02T00:00Z
    # a fixed zone offset shifts the local calendar day: 2025-01-31T20:00Z is Feb 1 in UTC+05:30
    assert oracle.add_months(1738353600000, 1, 330 * 60000) == 1738353600000 + 28 * 86400000
    assert oracle.add_months(1738353600000, 1, 0) == 1738353600000 + 31 * 86400000   # Jan 31 + 1 month = "Feb 31" = Mar 3


def test_g5_reference_trace_on_the_oracle(oracle):
    """1 345 recorded calls into the real sessionStore.js (create / get / touch / delete / delete-by-user / purge / census,
    same-millisecond sessions, lookups after death) replayed on numpy columns with the oracle's predicates: every answer
    the reference gave, and at every census the per-user feeds of the batched scan."""
    from trace_replay import OracleTable, load_trace, replay
    trace = load_trace(GOLDEN)
    assert replay(trace, OracleTable(oracle, len(trace["users"]))) == len(trace["ops"]) > 1300


def test_calendar_cutoff_matches_js_date_vectors(oracle):
    """a11, the window scalar (/root/reference/server/calendarFeed.js:33-38): the Python restatement (zoneinfo) against
    4 634 vectors produced by the JS engine's own Date under seven real time zones (tests/golden/cutoff_zones.json,
    oracle/gen_cutoff_golden.js): month-end overflow, leap years, DST change days of both hemispheres, a zone whose
    midnight does not exist on the change day (the hour the Date then reads is kept by setMonth), a zone that skipped a
    calendar day (Pacific/Apia, 2011-12-30).  The reference function itself cannot be imported here (node-ical, Node >= 14):
    the pin is on the engine semantics its three Date calls rely on."""
    doc = json.load(open(os.path.join(GOLDEN, "cutoff_zones.json")))
    assert len(doc["zones"]) == 7
    n = 0
    for tz, cases in doc["zones"].items():
        for now, back, want in cases:
            assert oracle.calendar_cutoff(now, back, tz) == want, (tz, now, back)
            n += 1
    assert n >= 4000
    # the hand-derived H4 cases agree with the engine-made ones
    h = json.load(open(os.path.join(GOLDEN, "hand_derived_h1_h5.json")))
    for c in h["cutoff_cases_tz_utc"]:
        assert oracle.calendar_cutoff(c["now_ms"], c["months_back"], "UTC") == c["expect_ms"]


def test_multithread_oracle_equals_single_thread(oracle):
    """bench.py's cpu_baseline B2 leg: pie_oracle_scan_mt (rows split over threads, per-user prefix over the threads, buckets
    ordered in parallel) gives pie_oracle_scan's bytes; pie_oracle_gen_mt the generator's."""
    n, U, D = 300007, 977, 32
    cols = oracle.gen(0x5EED5EED, n, 0, n, U, D, 1)
    for a, b in zip(cols, oracle.gen_mt(0x5EED5EED, n, 0, n, U, D, 1, threads=5)):
        assert np.array_equal(a, b)
    T0 = oracle.T0_MS
    for now, cutoff, mask in [(T0 - 6 * 3600 * 1000, T0 - 61 * 86400 * 1000, 0x55555555), (INT64_MIN, INT64_MIN, ALL), (2 ** 62, INT64_MIN, ALL)]:
        want = oracle.scan(*cols, U, now, cutoff, mask & 0xFFFFFFFF)
        for threads in (1, 3, 8):
            got = oracle.scan_mt(*cols, U, now, cutoff, mask & 0xFFFFFFFF, threads)
            for x, y in zip(got, want):
                assert np.array_equal(x, y)


def test_add_months_under_real_time_zones(oracle):
    """f2 under daylight saving: the oracle's restatement of `setMonth` on a LOCAL Date — offset at the UTC
    instant, civil month shift, skipped / repeated local times read with the offset before the transition — against the JS
    engine's own Date under seven zones (tests/golden/addmonths_zones.json, oracle/gen_addmonths_zones_golden.js), computed
    with the transition table the product's host built in the same JS process."""
    g = json.load(open(os.path.join(GOLDEN, "addmonths_zones.json")))
    n = 0
    for zone, z in g["zones"].items():
        T, off = z["transitions"], z["offsets"]
        assert len(off) == len(T) + 1
        for ts, m, want in z["cases"]:
            assert oracle.add_months_tz(ts, m, T, off) == want, (zone, ts, m)
            n += 1
        if not T:   # a zone without transitions is the fixed-offset form
            for ts, m, want in z["cases"][:50]:
                assert oracle.add_months(ts, m, off[0]) == want
    assert n > 10000


def test_zoneinfo_table_agrees_with_the_js_table(oracle):
    """The same table from a different engine and data set (Python zoneinfo): every transition the JS host found between 1990 and
    2037 is there, to the millisecond, with the same offsets — the table format and the probing are not an artefact of one engine."""
    g = json.load(open(os.path.join(GOLDEN, "addmonths_zones.json")))
    lo, hi = 631152000000, 2145916800000
    for zone in ("America/New_York", "Europe/Berlin", "Australia/Lord_Howe", "Asia/Kolkata"):
        T, off = oracle.tz_table(zone, lo, hi)
        z = g["zones"][zone]
        js = [(t, o) for t, o in zip(z["transitions"], z["offsets"][1:]) if lo < t <= hi]
        assert js == list(zip(T.tolist(), off[1:].tolist())), zone
