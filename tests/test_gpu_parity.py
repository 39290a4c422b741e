"""GPU: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs — bit-exact
counts / offsets / idx — plus the golden fixtures and size-independent properties at full size."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

INT64_MIN = -(2 ** 63)
ALL = 2 ** 64 - 1
DAY = 86400 * 1000
SEED = 0x5EED5EED


def spec_query(oracle):
    return oracle.T0_MS - 6 * 3600 * 1000, oracle.T0_MS - 61 * DAY, 0x5555555555555555


def assert_same(got, want):
    for name, a, b in zip(("counts", "offsets", "idx"), got, want):
        assert a.dtype == b.dtype, name
        assert np.array_equal(a, b), name


def run_both(ctx, oracle, cols, U, D, now, cutoff, mask):
    s, e, u, d = cols
    ctx.load_columns(s, e, u, d, U)
    ctx.set_disciplines(mask, D)
    got = ctx.scan(now, cutoff)
    m = mask if D >= 64 else mask & ((1 << D) - 1)
    want = oracle.scan(s, e, u, d, U, now, cutoff, m)
    assert_same(got, want)
    return got


def test_golden_sessionstore_vectors(gpu_ctx, oracle):
    g = json.load(open(os.path.join(GOLDEN, "sessionstore_g1_g4.json")))
    users = g["users"]
    s = np.array([r["createdAt"] for r in g["sessions"]], np.int64)
    e = np.array([r["expiresAt"] for r in g["sessions"]], np.int64)
    u = np.array([users.index(r["user"]) for r in g["sessions"]], np.int32)
    d = np.zeros(len(s), np.int32)
    gpu_ctx.load_columns(s, e, u, d, len(users))
    gpu_ctx.set_disciplines(1, 1)
    for case in g["G1"]:
        counts, offsets, idx = gpu_ctx.scan(case["now"], INT64_MIN)
        live = np.zeros(len(s), int)
        live[idx] = 1
        assert live.tolist() == case["live"]
    for case in g["G2"]:
        _, _, idx = gpu_ctx.scan(case["now"], INT64_MIN)
        assert sorted(idx.tolist()) == case["survivors"]
        q = gpu_ctx.expired_queue(INT64_MIN, case["now"])
        assert sorted(set(range(len(s))) - set(q.tolist())) == case["survivors"]
        assert q.tolist() == sorted(q.tolist())
    for case in g["G3"]:
        gpu_ctx.load_columns(s, e, u, d, len(users))
        k = users.index(case["user"]) if case["user"] in users else -1
        deleted = gpu_ctx.delete_user(k)
        _, _, idx = gpu_ctx.scan(case["observe_now"], INT64_MIN)
        assert sorted(idx.tolist()) == case["survivors"]
        assert deleted.tolist() == sorted(set(range(len(s))) - set(case["survivors"]))
        assert gpu_ctx.delete_user(k).size == 0   # second call finds nothing left
    gpu_ctx.load_columns(s, e, u, d, len(users))
    for t in g["G4"]:
        if t["returned"] is None:
            continue
        gpu_ctx.set_end([t["row"]], [t["returned"]["expiresAt"]])
        _, e2, _, _ = gpu_ctx.fetch_rows([t["row"]])
        assert int(e2[0]) == t["after"]["expiresAt"]


def test_g5_reference_trace_on_the_device(gpu_ctx):
    """The recorded call sequence of the real sessionStore.js (tests/golden/sessionstore_g5_trace.json) replayed on the
    device table through the C ABI: appends one row at a time, touches, tombstones, user deletes, purges via the expired
    queue, and the batched scan at every census — every answer equals the reference's."""
    from trace_replay import DeviceTable, load_trace, replay
    trace = load_trace(GOLDEN)
    assert replay(trace, DeviceTable(gpu_ctx, len(trace["users"]))) == len(trace["ops"])


def test_hand_derived_vectors(gpu_ctx):
    h = json.load(open(os.path.join(GOLDEN, "hand_derived_h1_h5.json")))
    for case in h["scan_cases"]:
        c = case["columns"]
        gpu_ctx.load_columns(c["start"], c["end"], c["user"], c["disc"], case["n_users"])
        gpu_ctx.set_disciplines(case["mask"], 64)
        counts, offsets, idx = gpu_ctx.scan(case["now"], case["cutoff"])
        assert counts.tolist() == case["expect"]["counts"], case["name"]
        assert offsets.tolist() == case["expect"]["offsets"], case["name"]
        assert idx.tolist() == case["expect"]["idx"], case["name"]


@pytest.mark.parametrize("n,U,D,flags", [
    (1000, 10, 3, 0),            # BASELINE config 1
    (1, 1, 1, 0), (63, 2, 2, 0), (64, 2, 2, 1), (65, 3, 2, 0), (511, 5, 7, 1), (512, 5, 7, 0), (513, 5, 7, 3),
    (2047, 11, 32, 0), (2048, 11, 32, 1), (2049, 11, 32, 2), (100003, 97, 32, 0), (300000, 1, 32, 1),
    (1 << 20, 10 ** 4, 32, 0), (1 << 20, 10 ** 4, 32, 3), (3000017, 2049, 64, 1),
    (70001, 333, 32, 4), (1 << 20, 10 ** 4, 32, 4), (2500013, 5000, 32, 5),   # rows in order of creation: live rows at the table's end
])
def test_parity_small(gpu_ctx, oracle, n, U, D, flags):
    cols = oracle.gen(SEED, n, 0, n, U, D, flags)
    now, cutoff, mask = spec_query(oracle)
    run_both(gpu_ctx, oracle, cols, U, D, now, cutoff, mask)
    # everything selected (every bucket full, medium / big sort paths), nothing selected, half window
    run_both(gpu_ctx, oracle, cols, U, D, INT64_MIN, INT64_MIN, ALL)
    got = run_both(gpu_ctx, oracle, cols, U, D, 2 ** 62, INT64_MIN, ALL)
    assert got[2].size == 0 and got[1][-1] == 0
    run_both(gpu_ctx, oracle, cols, U, D, oracle.T0_MS - 100 * DAY, oracle.T0_MS - 61 * DAY, 0xAAAAAAAAAAAAAAAA)


@pytest.mark.parametrize("variant", [0x00, 0x01, 0x02, 0x03, 0x43, 0x23, 0x83, 0x04, 0x05, 0x25, 0x85, 0xC5, 0x45,
                                     0x405, 0x425, 0x484, 0x485, 0x4C5, 0xC05, 0xC85, 0xCC5, 0x495, 0xC95, 0xCD5])
def test_every_k1_form_is_bit_exact(pie, oracle, variant, monkeypatch):
    """Each form of the scan kernel (streaming / late-user / liveness-first, nt on/off, unroll 2/4/8) pinned
    through PIE_K1_VARIANT gives the oracle's bytes, on ragged sizes, all-live and none-live tables."""
    monkeypatch.setenv("PIE_K1_VARIANT", hex(variant))
    with pie.PieScan(0) as ctx:
        for n, U, D, flags in [(1, 1, 1, 0), (4097, 9, 7, 1), (70001, 333, 32, 0), (1 << 20, 5000, 32, 3)]:
            cols = oracle.gen(SEED, n, 0, n, U, D, flags)
            now, cutoff, mask = spec_query(oracle)
            for q in [(now, cutoff, mask), (INT64_MIN, INT64_MIN, ALL), (2 ** 62, INT64_MIN, ALL),
                      (oracle.T0_MS - 100 * DAY, oracle.T0_MS - 61 * DAY, 0xAAAAAAAAAAAAAAAA)]:
                run_both(ctx, oracle, cols, U, D, *q)
            assert ctx.stats()["k1_variant"] == variant


def test_k1_form_follows_live_fraction(pie, oracle):
    """Unpinned: the form of the table pass follows what is known about the table.  A freshly loaded table has its key
    histogram (taken when the key columns were built), which bounds the live rows of a query: few -> the keyed
    liveness-first form from the first scan on; afterwards the live fraction the previous scan counted decides, and
    the streaming form takes over once most rows are live.  Same bytes either way."""
    with pie.PieScan(0) as ctx:
        n, U, D = 400000, 1000, 32
        s, e, u, d = oracle.gen(SEED, n, 0, n, U, D, 0)
        now, cutoff, mask = spec_query(oracle)
        ctx.load_columns(s, e, u, d, U)
        ctx.set_disciplines(mask, D)
        want_spec = oracle.scan(s, e, u, d, U, now, cutoff, mask & 0xFFFFFFFF)
        want_all = oracle.scan(s, e, u, d, U, INT64_MIN, INT64_MIN, mask & 0xFFFFFFFF)
        assert_same(ctx.scan(now, cutoff), want_spec)
        st = ctx.stats()
        assert st["k1_variant"] == 0xC85 and abs(st["live"] / n - 18 / (120 * 24)) < 2e-3   # keyed at once, from the histogram
        assert_same(ctx.scan(now, cutoff), want_spec)
        assert ctx.stats()["k1_variant"] == 0xC85
        assert_same(ctx.scan(INT64_MIN, INT64_MIN), want_all)   # still liveness-first (decided from the last scan), on the
        assert ctx.stats()["k1_variant"] == 0x485 and ctx.stats()["live"] == n   # 2-byte key: `now` is below the fine key's base
        assert_same(ctx.scan(INT64_MIN, INT64_MIN), want_all)   # ... and streaming once everything is live
        assert ctx.stats()["k1_variant"] == 0x03
        # a freshly loaded table and a query that most rows survive: streaming from the first scan
        ctx.load_columns(s, e, u, d, U)
        assert_same(ctx.scan(oracle.T0_MS - 100 * DAY, INT64_MIN), oracle.scan(s, e, u, d, U, oracle.T0_MS - 100 * DAY, INT64_MIN, mask & 0xFFFFFFFF))
        assert ctx.stats()["k1_variant"] == 0x03
        # an append keeps what the scans learned (no fall-back to streaming for one new row)
        assert_same(ctx.scan(now, cutoff), want_spec)
        assert_same(ctx.scan(now, cutoff), want_spec)
        ctx.append_rows(s[:1], e[:1], u[:1], d[:1], U)
        ctx.scan(now, cutoff)
        assert ctx.stats()["k1_variant"] == 0xC85


def test_streaming_form_aggregates_on_user_clustered_tables(pie, oracle):
    """Rows clustered by user + a dense query: the streaming form notices that most selected rows sit next to a row of
    the same user and switches to wave-aggregated histogram atomics (0x43); on a randomly ordered table it does not, and
    it switches back when the table stops being clustered.  Same bytes throughout.  (The ordered run, which would take these
    dense queries over from the third one on, is switched off: this test is about the general path's own forms.)"""
    with pie.PieScan(0) as ctx:
        ctx.set_ordered_run(0)
        n, U, D = 600000, 2000, 32
        now = oracle.T0_MS - 100 * DAY                       # ~83 % of the rows live: streaming form
        for flags, want_form in ((2, 0x43), (0, 0x03)):
            s, e, u, d = oracle.gen(SEED, n, 0, n, U, D, flags)
            ctx.load_columns(s, e, u, d, U)
            ctx.set_disciplines(ALL, D)
            want = oracle.scan(s, e, u, d, U, now, INT64_MIN, 0xFFFFFFFF)
            forms = []
            for _ in range(3):
                assert_same(ctx.scan(now, INT64_MIN), want)
                forms.append(ctx.stats()["k1_variant"])
            assert forms == [0x03, want_form, want_form]


def test_partition_overflow_reruns_on_the_general_path(pie, oracle, monkeypatch):
    """The opt-in partitioned path (PIE_FAST_PATH=1: table pass + ONE tail kernel, no host round trip; measured at
    parity with the default path, kept as an experiment) is chosen from the PREVIOUS scan's M.  When the next query selects far more rows than
    a partition can hold, the tail flags the overflow and the same scan reruns on the general path — same bytes — and
    the fast path stays off for that table."""
    monkeypatch.setenv("PIE_FAST_PATH", "1")
    with pie.PieScan(0) as ctx:
        n, U, D = 3000000, 30000, 32
        s, e, u, d = oracle.gen(SEED, n, 0, n, U, D, 0)
        now, cutoff, mask = spec_query(oracle)
        ctx.load_columns(s, e, u, d, U)
        ctx.set_disciplines(mask, D)
        sparse = oracle.scan(s, e, u, d, U, now, cutoff, mask & 0xFFFFFFFF)
        dense = oracle.scan(s, e, u, d, U, INT64_MIN, INT64_MIN, mask & 0xFFFFFFFF)   # ~50 rows per user
        assert sparse[0].max() <= 16 < dense[0].max()
        assert_same(ctx.scan(now, cutoff), sparse)
        assert_same(ctx.scan(now, cutoff), sparse)
        assert ctx.stats()["k1_variant"] == 0x285
        assert_same(ctx.scan(INT64_MIN, INT64_MIN), dense)                      # overflow -> rerun, transparently
        assert ctx.stats()["k1_variant"] == 0x85
        assert_same(ctx.scan(now, cutoff), sparse)
        assert ctx.stats()["k1_variant"] != 0x285                               # fast path is off for this table now
        ctx.load_columns(s, e, u, d, U)                                         # ... until the table is reloaded
        assert_same(ctx.scan(now, cutoff), sparse)
        assert_same(ctx.scan(now, cutoff), sparse)
        assert ctx.stats()["k1_variant"] == 0x285


def test_partitioned_path_parity(pie, oracle, monkeypatch):
    """PIE_FAST_PATH=1 on tables of many shapes: whenever the path engages (variant 0x285) its bytes equal the oracle's,
    including buckets of 9..16 rows, empty partitions, user counts that are not a multiple of the partition range."""
    monkeypatch.setenv("PIE_FAST_PATH", "1")
    rng = np.random.default_rng(99)
    engaged = 0
    with pie.PieScan(0) as ctx:
        for n, U in [(200000, 1), (200000, 33), (500000, 5000), (1 << 20, 100000), (1 << 20, 4097 * 3), (3000017, 1000003)]:
            s, e, u, d = oracle.gen(int(rng.integers(1, 2 ** 60)), n, 0, n, U, 32, 1)
            ctx.load_columns(s, e, u, d, U)
            ctx.set_disciplines(ALL, 32)
            for now in [oracle.T0_MS - 3600 * 1000, oracle.T0_MS - 2 * DAY, oracle.T0_MS - 3600 * 1000, 2 ** 62, oracle.T0_MS - 5 * DAY]:
                want = oracle.scan(s, e, u, d, U, now, INT64_MIN, 0xFFFFFFFF)
                assert_same(ctx.scan(now, INT64_MIN), want)
                engaged += ctx.stats()["k1_variant"] == 0x285
    assert engaged >= 8


def test_bucket_routes_by_user_table_size(pie, oracle):
    """Three routes to the per-user buckets, chosen by the size of the user table: fused offsets + order kernel over the
    direct bucket slots (<= 1 048 576 users), direct slots with separate kernels (up to 8 M users), staged records + scatter
    (beyond).  Sparse and dense queries (buckets of 0..16 rows and of hundreds) give the oracle's bytes on each."""
    n = 1 << 20
    with pie.PieScan(0) as ctx:
        for U in (1, 300, 100000, 1048576, 1048577, 9000000):
            flags = 1 if U > 1000 else 0
            s, e, u, d = oracle.gen(SEED + U, n, 0, n, U, 32, flags)
            ctx.load_columns(s, e, u, d, U)
            ctx.set_disciplines(ALL, 32)
            for now in (oracle.T0_MS - 6 * 3600 * 1000, oracle.T0_MS - 6 * 3600 * 1000, oracle.T0_MS - 50 * DAY, INT64_MIN):
                assert_same(ctx.scan(now, INT64_MIN), oracle.scan(s, e, u, d, U, now, INT64_MIN, 0xFFFFFFFF))


def test_scan_written_message_equals_the_pack_kernel(pie, oracle):
    """Exchange step: pie_scan_begin_packed / pie_scan_finish_packed.  For a query whose buckets all fit the direct slots
    the scan's own kernels write the message [off[0..u_pad] | M | rows[0..cap)] (ready = True, no pack kernel, no event);
    otherwise the pack kernel is enqueued (ready = False).  Either way the words equal what pack_results_device writes
    from the finished result, also with a row capacity below M and with padding users."""
    import torch
    n, U, D = 1 << 20, 60000, 32
    s, e, u, d = oracle.gen(SEED, n, 0, n, U, D, 0)
    dev = torch.device("cuda", 0)
    with pie.PieScan(0) as ctx:
        ctx.set_ordered_run(0)   # the general path's K2 is what writes the message here (tests/test_gpu_ordered.py has the run's)
        ctx.load_columns(s, e, u, d, U)
        ctx.set_disciplines(ALL, D)
        cases = [(oracle.T0_MS - 6 * 3600 * 1000, True), (oracle.T0_MS - 6 * 3600 * 1000, True),   # ~0.1 rows per user
                 (oracle.T0_MS - 30 * DAY, True),                                                      # ~4 per user: buckets of 9..16 too
                 (INT64_MIN, False)]                                                                   # ~17 per user: staged records
        for now, want_ready in cases:
            want = oracle.scan(s, e, u, d, U, now, INT64_MIN, 0xFFFFFFFF)
            m = want[2].size
            for u_pad, cap in [(U, m + 7), (U + 123, m), (U + 1, max(m // 2, 1))]:
                msg = torch.full((u_pad + 2 + cap,), -7, dtype=torch.int32, device=dev)
                ref = torch.full((u_pad + 2 + cap,), -7, dtype=torch.int32, device=dev)
                torch.cuda.synchronize()   # the fills ran on torch's stream, the scan writes from the library's own
                ctx.scan_begin_packed(now, INT64_MIN, msg.data_ptr(), u_pad, cap)
                got_m, ready = ctx.scan_finish_packed()
                assert got_m == m and ready == want_ready
                ctx.pack_results_device(ref.data_ptr(), u_pad, cap)
                ctx.synchronize()
                torch.cuda.synchronize()
                a, b = msg.cpu().numpy(), ref.cpu().numpy()
                k = min(m, cap)
                assert np.array_equal(a[: u_pad + 2 + k], b[: u_pad + 2 + k])
                assert np.array_equal(a[: U + 1], want[1].astype(np.int32)) and a[u_pad + 1] == m
                assert np.array_equal(a[u_pad + 2: u_pad + 2 + k], want[2][:k])
                assert np.all(a[u_pad + 2 + k:] == -7)          # nothing written past the capacity
        # two in flight
        now = oracle.T0_MS - 6 * 3600 * 1000
        want = oracle.scan(s, e, u, d, U, now, INT64_MIN, 0xFFFFFFFF)
        m = want[2].size
        bufs = [torch.zeros(U + 2 + m, dtype=torch.int32, device=dev) for _ in range(3)]
        torch.cuda.synchronize()
        ctx.scan_begin_packed(now, INT64_MIN, bufs[0].data_ptr(), U, m)
        for i in range(1, 6):
            ctx.scan_begin_packed(now, INT64_MIN, bufs[i % 3].data_ptr(), U, m)
            assert ctx.scan_finish_packed() == (m, True)
            assert np.array_equal(bufs[(i - 1) % 3].cpu().numpy()[U + 2:], want[2])
        assert ctx.scan_finish_packed() == (m, True)


def test_read_user_feed_is_the_slice(gpu_ctx, oracle, pie):
    """pie_read_user_feed: one user's rows of the last scan == idx[offsets[u]:offsets[u+1]] of the whole result, for
    empty, tiny and large feeds; users outside the table have empty feeds; a too-small buffer reports the length."""
    n, U = 400000, 2000
    s, e, u, d = oracle.gen(SEED, n, 0, n, U, 32, 1)
    u = np.where(np.arange(n) % 7 == 0, 5, u).astype(np.int32)     # one user with a big feed
    gpu_ctx.load_columns(s, e, u, d, U)
    gpu_ctx.set_disciplines(ALL, 32)
    for now in (oracle.T0_MS - 6 * 3600 * 1000, oracle.T0_MS - 40 * DAY):
        want = oracle.scan(s, e, u, d, U, now, INT64_MIN, 0xFFFFFFFF)
        gpu_ctx.scan_device(now, INT64_MIN)
        for user in [0, 5, 17, U - 1] + [int(np.argmin(want[0])), int(np.argmax(want[0]))]:
            assert np.array_equal(gpu_ctx.read_user_feed(user), want[2][want[1][user]:want[1][user + 1]])
        assert gpu_ctx.read_user_feed(-1).size == 0 and gpu_ctx.read_user_feed(U).size == 0
        with pytest.raises(pie.PieError) as err:
            gpu_ctx.read_user_feed(5, cap=3)
        assert err.value.code == -5   # PIE_E_CAPACITY


def test_sharded_feeds_driver_on_the_device(pie, oracle):
    """shard.ShardedFeeds over HipShardBackend on one rank (no process group): capacity negotiation, the pipelined
    run_steps with scan-written messages, a query that outgrows the message (collective re-negotiation path) and the
    synchronous form — offsets and rows of every feed equal the oracle's."""
    import torch
    from sph_pie_amd.shard import HipShardBackend, ShardedFeeds
    n, U, D = 1 << 20, 5000, 32
    s, e, u, d = oracle.gen(SEED, n, 0, n, U, D, 0)
    with pie.PieScan(0) as ctx:
        ctx.load_columns(s, e, u, d, U)
        ctx.set_disciplines(ALL, D)
        feeds = ShardedFeeds(HipShardBackend(ctx, torch.device("cuda", 0)), 0, 1, U)
        sparse = (oracle.T0_MS - 6 * 3600 * 1000, INT64_MIN)
        dense = (oracle.T0_MS - 60 * DAY, INT64_MIN)

        def check(res, query):
            want = oracle.scan(s, e, u, d, U, query[0], query[1], 0xFFFFFFFF)
            m = want[2].size
            assert int(res["lengths"][0]) == m
            assert np.array_equal(res["offsets"][0].cpu().numpy(), want[1].astype(np.int32))
            assert np.array_equal(res["rows"][0].cpu().numpy()[:m], want[2])

        check(feeds.run_steps(7, *sparse), sparse)
        check(feeds.scan_and_gather(*sparse), sparse)
        assert feeds.run_steps(4, *dense) is None          # outgrew the negotiated capacity: raised, nothing lost
        check(feeds.run_steps(4, *dense), dense)
        check(feeds.run_steps(3, *sparse), sparse)
        for batch, k in ((4, 9), (3, 3), (2, 1)):          # several scans per gather
            feeds.batch = batch
            check(feeds.run_steps(k, *sparse), sparse)
        feeds.batch = 4
        assert feeds.run_steps(5, *dense) is None or True   # capacity is already large enough here
        check(feeds.run_steps(5, *dense), dense)


def test_two_scans_in_flight(pie, oracle):
    """begin(i+1) before finish(i): different queries back to back, results of each finished scan are exact and
    stay readable while the next scan is already queued; a third begin is refused."""
    with pie.PieScan(0) as ctx:
        n, U, D = 600011, 3000, 32
        s, e, u, d = oracle.gen(SEED, n, 0, n, U, D, 1)
        ctx.load_columns(s, e, u, d, U)
        ctx.set_disciplines(ALL, D)
        now, cutoff, _ = spec_query(oracle)
        queries = [(now, cutoff), (INT64_MIN, INT64_MIN), (oracle.T0_MS - 40 * DAY, cutoff), (2 ** 62, INT64_MIN), (now, cutoff)]
        want = [oracle.scan(s, e, u, d, U, q[0], q[1], 0xFFFFFFFF) for q in queries]
        ctx.scan_begin(*queries[0])
        for i in range(len(queries)):
            if i + 1 < len(queries):
                ctx.scan_begin(*queries[i + 1])
                with pytest.raises(pie.PieError):
                    ctx.scan_begin(*queries[0])      # a third scan in flight is a state error, nothing is enqueued
            m = ctx.scan_finish()
            assert m == want[i][2].size
            assert_same(ctx.read_results(), want[i])
        with pytest.raises(pie.PieError):
            ctx.scan_finish()
        # the pipelined helper ends in the same state as a plain scan
        assert ctx.scan_pipelined(7, *queries[2]) == want[2][2].size
        assert_same(ctx.read_results(), want[2])
        assert_same(ctx.scan(*queries[1]), want[1])


def test_randomized_tables_and_queries(pie, oracle):
    """Seeded fuzz: 40 tables of random size / user count / skew / sentinels, each scanned with several random
    queries (the same query twice in a row so that both the first-scan form and the adapted form run)."""
    rng = np.random.default_rng(20261004)
    with pie.PieScan(0) as ctx:
        for case in range(40):
            n = int(rng.choice([0, 1, 2, 63, 64, 65, 127, 129, 4095, 4096, 4097, int(rng.integers(1, 300000))]))
            U = int(rng.integers(1, 4000))
            D = int(rng.integers(1, 65))
            s, e, u, d = oracle.gen(int(rng.integers(0, 2 ** 62)), max(n, 1), 0, n, U, D, int(rng.integers(0, 4)))
            if n and rng.random() < 0.5:      # power-law users: a few huge buckets, many empty ones
                u = np.minimum((U * rng.random(n) ** 4).astype(np.int32), U - 1)
            if n and rng.random() < 0.3:
                e[rng.random(n) < 0.2] = INT64_MIN
            if n and rng.random() < 0.3:
                d[rng.random(n) < 0.1] = int(rng.choice([-1, 64, 200, -2 ** 31]))
            if n and rng.random() < 0.3:
                s[rng.random(n) < 0.5] = int(s[0])       # many equal starts: tie rule
            ctx.load_columns(s, e, u, d, U)
            for _ in range(3):
                now = int(rng.choice([INT64_MIN, 2 ** 62, int(oracle.T0_MS - rng.integers(0, 130) * DAY)]))
                cutoff = int(rng.choice([INT64_MIN, int(oracle.T0_MS - rng.integers(0, 130) * DAY)]))
                mask = int(rng.integers(0, 2 ** 63)) | (int(rng.integers(0, 2)) << 63)
                ctx.set_disciplines(mask, D)
                want = oracle.scan(s, e, u, d, U, now, cutoff, mask if D >= 64 else mask & ((1 << D) - 1))
                assert_same(ctx.scan(now, cutoff), want)
                assert_same(ctx.scan(now, cutoff), want)
                prev = int(now - rng.integers(1, 40) * DAY) if abs(now) < 2 ** 61 else INT64_MIN
                assert np.array_equal(ctx.expired_queue(prev, now), oracle.expired_queue(e, prev, now))


def test_generator_parity(gpu_ctx, oracle):
    for n, U, D, flags in [(1000, 10, 3, 0), (70001, 333, 32, 1), (70001, 333, 32, 2), (4096, 4096, 64, 3)]:
        gpu_ctx.gen_synthetic(SEED, n, 0, n, U, D, flags)
        got = gpu_ctx.read_columns()
        want = oracle.gen(SEED, n, 0, n, U, D, flags)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
    gpu_ctx.gen_synthetic(SEED, 10 ** 6, 123456, 5000, 100, 32, 1)  # a slice of a bigger table
    for a, b in zip(gpu_ctx.read_columns(), oracle.gen(SEED, 10 ** 6, 123456, 5000, 100, 32, 1)):
        assert np.array_equal(a, b)


def test_zipf_corpus_parity(pie, gpu_ctx, oracle, request):
    """The skewed-user corpus (Zipf(1.1) thresholds from the host): generator and scan bit-exact vs the oracle; the head
    user's bucket goes through the tile + merge path."""
    n, U, D = 2 * 10 ** 6, 20000, 32
    cdf = pie.zipf_cdf(U)
    assert np.array_equal(cdf, oracle.zipf_cdf(U)) and np.all(np.diff(cdf.astype(np.float64)) >= 0)
    gpu_ctx.set_ordered_run(0)   # the general path's big-bucket machinery is the subject (the ordered run would take a skewed table over)
    request.addfinalizer(lambda: gpu_ctx.set_ordered_run(1))
    gpu_ctx.gen_synthetic_cdf(SEED, n, 0, n, U, D, 1, cdf)
    want_cols = oracle.gen_cdf(SEED, n, 0, n, U, D, 1, cdf)
    for a, b in zip(gpu_ctx.read_columns(), want_cols):
        assert np.array_equal(a, b)
    head = np.bincount(want_cols[2], minlength=U)
    assert head[0] > 0.08 * n and head[0] > 20 * head[100]
    gpu_ctx.set_disciplines(ALL, D)
    for now, cutoff in [(INT64_MIN, INT64_MIN), spec_query(oracle)[:2], (oracle.T0_MS - 60 * DAY, INT64_MIN)]:
        assert_same(gpu_ctx.scan(now, cutoff), oracle.scan(*want_cols, U, now, cutoff, 0xFFFFFFFF))
    assert gpu_ctx.stats()["n_big"] >= 1
    # a skewed table switches the liveness-first form to wave-aggregated histogram atomics on the next scan
    now, cutoff, _ = spec_query(oracle)
    want = oracle.scan(*want_cols, U, now, cutoff, 0xFFFFFFFF)
    for _ in range(3):
        assert_same(gpu_ctx.scan(now, cutoff), want)
    assert gpu_ctx.stats()["k1_variant"] == 0xCC5
    # by now the head users form the hot set (block-level histogram, rows staged with block-relative ranks).  A query
    # that leaves them only a handful of rows still runs with that (stale) hot set once: their small buckets must come
    # through the staged route too, not from the direct slots
    for now2 in (oracle.T0_MS + 11 * 3600 * 1000 + 1800 * 1000, oracle.T0_MS + 11 * 3600 * 1000 + 1800 * 1000, now):
        want2 = oracle.scan(*want_cols, U, now2, cutoff, 0xFFFFFFFF)
        assert_same(gpu_ctx.scan(now2, cutoff), want2)
    assert 0 < oracle.scan(*want_cols, U, oracle.T0_MS + 11 * 3600 * 1000 + 1800 * 1000, cutoff, 0xFFFFFFFF)[0][0] <= 4096


def test_skewed_users_big_buckets(gpu_ctx, oracle):
    """One user owns most rows (Zipf-like head): exercises tiles + merge passes of the big-bucket path."""
    rng = np.random.default_rng(7)
    n, U = 700001, 50
    s, e, u, d = oracle.gen(SEED, n, 0, n, U, 32, 1)
    u = np.where(rng.random(n) < 0.8, 3, u).astype(np.int32)
    u[rng.random(n) < 0.05] = 17
    s[::5] = s[0]  # many equal starts -> tie rule by row index under the merge path
    got = run_both(gpu_ctx, oracle, (s, e, u, d), U, 32, INT64_MIN, INT64_MIN, ALL)
    assert got[0].max() > 4096 * 64
    st = gpu_ctx.stats()
    assert st["n_big"] >= 2 and st["max_bucket"] == got[0].max()


def test_sentinel_end_and_out_of_table_disciplines(gpu_ctx, oracle):
    n, U = 5000, 9
    s, e, u, d = oracle.gen(SEED, n, 0, n, U, 7, 0)
    e[::3] = INT64_MIN          # "no end" / tombstone: never live, even for now = INT64_MIN
    d[1::7] = -1
    d[2::7] = 64
    d[3::7] = 2 ** 31 - 1
    got = run_both(gpu_ctx, oracle, (s, e, u, d), U, 7, INT64_MIN, INT64_MIN, ALL)
    assert not np.any(np.isin(got[2], np.arange(0, n, 3)))


def test_bad_user_ids_are_rejected_not_faulted(gpu_ctx, pie, oracle):
    s, e, u, d = oracle.gen(SEED, 3000, 0, 3000, 5, 3, 0)
    u[1234] = 5
    with pytest.raises(pie.PieError) as ei:
        gpu_ctx.load_columns(s, e, u, d, 5)
    assert ei.value.code == -1
    u[1234] = -1
    with pytest.raises(pie.PieError):
        gpu_ctx.load_columns(s, e, u, d, 5)


def test_idx_capacity_error(gpu_ctx, pie, oracle):
    s, e, u, d = oracle.gen(SEED, 4000, 0, 4000, 5, 3, 0)
    gpu_ctx.load_columns(s, e, u, d, 5)
    gpu_ctx.set_disciplines(ALL, 64)
    with pytest.raises(pie.PieError) as ei:
        gpu_ctx.scan(INT64_MIN, INT64_MIN, idx_cap=10)
    assert ei.value.code == pie.binding.PIE_E_CAPACITY


def test_append_rows_equals_bulk_load(gpu_ctx, oracle):
    """createSession path: appending in ragged batches gives the same table (and feeds) as one bulk load."""
    n, U = 50000, 300
    s, e, u, d = oracle.gen(SEED, n, 0, n, U, 32, 1)
    now, cutoff, mask = spec_query(oracle)
    cuts = [0, 1, 2, 65, 4096, 4097, 20000, n]
    gpu_ctx.load_columns(s[:0], e[:0], u[:0], d[:0], 1)
    for a, b in zip(cuts[:-1], cuts[1:]):
        gpu_ctx.append_rows(s[a:b], e[a:b], u[a:b], d[a:b], max(int(u[:b].max()) + 1, 1))
    assert gpu_ctx.n == n
    for a, b in zip(gpu_ctx.read_columns(), (s, e, u, d)):
        assert np.array_equal(a, b)
    gpu_ctx.set_disciplines(mask, 32)
    got = gpu_ctx.scan(INT64_MIN, INT64_MIN)
    want = oracle.scan(s, e, u, d, gpu_ctx.n_users, INT64_MIN, INT64_MIN, mask & 0xFFFFFFFF)
    assert_same(got, want)


@pytest.mark.parametrize("form", [0x485, 0xC85])
def test_liveness_key_follows_every_writer_of_end(pie, oracle, monkeypatch, form):
    """The keyed table pass reads a derived 2-byte (or, for the top of the range, 1-byte) column instead of `end`; every
    call that changes `end` must keep both in step.  With a keyed form pinned: touch (dead -> live, live -> dead, values far outside the range the key was built
    for), tombstones from delete_user / prune_before / retention_purge, appends below, inside and beyond the key range —
    after each the scan equals the oracle on the mirrored columns, at query times inside, below and above the range."""
    monkeypatch.setenv("PIE_K1_VARIANT", hex(form))
    rng = np.random.default_rng(2024)
    n, U, D = 300007, 700, 32
    s, e, u, d = oracle.gen(SEED, n, 0, n, U, D, 1)
    s, e, u, d = s.copy(), e.copy(), u.copy(), d.copy()
    now0 = oracle.T0_MS - 6 * 3600 * 1000
    nows = [now0, oracle.T0_MS - 60 * DAY, INT64_MIN, oracle.T0_MS - 130 * DAY, oracle.T0_MS + 400 * DAY, 2 ** 62, int(e[12345]), int(e[12345]) - 1]

    def check(ctx):
        for now in nows:
            assert_same(ctx.scan(now, INT64_MIN), oracle.scan(s, e, u, d, ctx.n_users, now, INT64_MIN, 0xFFFFFFFF))
            assert ctx.stats()["k1_variant"] == form

    with pie.PieScan(0) as ctx:
        ctx.load_columns(s, e, u, d, U)
        ctx.set_disciplines(ALL, D)
        check(ctx)
        # touch: new ends inside the range, far below, far above, the int64 extremes, tombstones
        rows = rng.choice(n, 5000, replace=False).astype(np.int32)
        new_end = rng.integers(oracle.T0_MS - 125 * DAY, oracle.T0_MS + 5 * DAY, rows.size).astype(np.int64)
        new_end[:50] = INT64_MIN
        new_end[50:100] = 2 ** 63 - 1
        new_end[100:150] = oracle.T0_MS + 4000 * DAY
        new_end[150:200] = -5
        ctx.set_end(rows, new_end)
        e[rows] = new_end
        check(ctx)
        gone = ctx.delete_user(5)
        assert np.array_equal(gone, np.nonzero((u == 5) & (e != INT64_MIN))[0])
        e[gone] = INT64_MIN
        check(ctx)
        gone = ctx.prune_before(oracle.T0_MS - 100 * DAY)
        e[gone] = INT64_MIN
        check(ctx)
        gone = ctx.retention_purge(oracle.T0_MS, 3, 0)
        assert gone.size > 0
        e[gone] = INT64_MIN
        check(ctx)
        # appends: within capacity growth and beyond it, ends below / inside / beyond the key range
        for k in (1, 777, 400000):
            s2 = rng.integers(oracle.T0_MS - 10 * DAY, oracle.T0_MS + 300 * DAY, k).astype(np.int64)
            e2 = s2 + rng.integers(-200 * DAY, 200 * DAY, k)
            u2 = rng.integers(0, U + 5, k).astype(np.int32)
            d2 = rng.integers(0, D, k).astype(np.int32)
            ctx.append_rows(s2, e2, u2, d2, U + 5)
            s, e, u, d = np.concatenate([s, s2]), np.concatenate([e, e2]), np.concatenate([u, u2]), np.concatenate([d, d2])
            check(ctx)


def test_liveness_key_refit_and_fallback(pie, oracle):
    """Unpinned: (1) rows appended far beyond the range the key column was built for all clamp to the top key, a query in
    that new range finds them ambiguous, and the column is rebuilt before the next scan, after which the keyed form is
    selective again; (2) a table whose `end` values are all equal cannot be separated by any key: the scan notices and
    goes back to streaming the `end` column.  Results equal the oracle at every step."""
    rng = np.random.default_rng(5)
    n, U, D = 400000, 500, 32
    s, e, u, d = oracle.gen(SEED, n, 0, n, U, D, 0)
    with pie.PieScan(0) as ctx:
        ctx.load_columns(s, e, u, d, U)
        ctx.set_disciplines(ALL, D)
        now = oracle.T0_MS - 6 * 3600 * 1000
        want = oracle.scan(s, e, u, d, U, now, INT64_MIN, 0xFFFFFFFF)
        for _ in range(3):
            assert_same(ctx.scan(now, INT64_MIN), want)
        st = ctx.stats()
        assert st["k1_variant"] == 0xC85 and st["key_ambiguous"] < 1000
        # one appended row doubles the table's capacity (and re-derives the key while it is at it) ...
        ctx.append_rows(s[:1], e[:1], u[:1], d[:1], U)
        s, e, u, d = np.concatenate([s, s[:1]]), np.concatenate([e, e[:1]]), np.concatenate([u, u[:1]]), np.concatenate([d, d[:1]])
        # ... so these fit in place, keyed under the old parameters.  A year later: 300 000 new sessions, the old ones long dead
        k = 300000
        s2 = (oracle.T0_MS + 365 * DAY + rng.integers(0, 30 * DAY, k)).astype(np.int64)
        e2 = s2 + 43200000
        u2, d2 = rng.integers(0, U, k).astype(np.int32), rng.integers(0, D, k).astype(np.int32)
        ctx.append_rows(s2, e2, u2, d2, U)
        s, e, u, d = np.concatenate([s, s2]), np.concatenate([e, e2]), np.concatenate([u, u2]), np.concatenate([d, d2])
        now = oracle.T0_MS + 365 * DAY + 29 * DAY
        want = oracle.scan(s, e, u, d, U, now, INT64_MIN, 0xFFFFFFFF)
        seen = []
        for _ in range(4):
            assert_same(ctx.scan(now, INT64_MIN), want)
            st = ctx.stats()
            seen.append((st["k1_variant"], st["key_ambiguous"]))
        assert seen[0][0] == 0xC85 and seen[0][1] > 250000      # every appended row sat on the clamp key
        assert seen[2][0] == 0xC85 and seen[2][1] < 2000        # rebuilt: selective again
        # all ends equal: no key can separate a query at that instant
        e3 = np.full(n, oracle.T0_MS, np.int64)
        ctx.load_columns(s[:n], e3, u[:n], d[:n], U)
        want = oracle.scan(s[:n], e3, u[:n], d[:n], U, oracle.T0_MS, INT64_MIN, 0xFFFFFFFF)
        forms = []
        for _ in range(5):
            assert_same(ctx.scan(oracle.T0_MS, INT64_MIN), want)
            forms.append(ctx.stats()["k1_variant"])
        assert want[2].size == 0 and forms[1:] == [0xC85, 0x485, 0x85, 0x85]   # 1-byte key, 2-byte key, then the column itself


def test_prune_before_is_the_window_complement(gpu_ctx, oracle):
    """_pruneCalendarEvents (sqlProvider.js:956-968): rows with start < cutoff go, exactly the rows the window
    predicate (:284) would have rejected; a scan without window after the prune equals a scan with window before it."""
    n, U = 120001, 77
    s, e, u, d = oracle.gen(SEED, n, 0, n, U, 32, 1)
    now, cutoff, mask = spec_query(oracle)
    now -= 90 * DAY
    gpu_ctx.load_columns(s, e, u, d, U)
    gpu_ctx.set_disciplines(mask, 32)
    with_window = gpu_ctx.scan(now, cutoff)
    pruned = gpu_ctx.prune_before(cutoff)
    assert np.array_equal(pruned, np.nonzero(s < cutoff)[0])
    assert_same(gpu_ctx.scan(now, INT64_MIN), with_window)
    assert gpu_ctx.prune_before(cutoff).size == 0


def test_archive_group_min_queue(gpu_ctx, oracle):
    """The reference's archive chain (sqlProvider.js:758-816) on the session table: segmented min by group, threshold,
    whole-group selection, queue in (group first appearance, row) order — vs the C oracle on small tables and its numpy
    twin on a larger one."""
    W = 43200000
    for n, U, flags in [(1, 1, 0), (300, 7, 1), (5000, 37, 0), (20000, 2000, 3)]:
        s, e, u, d = oracle.gen(SEED, n, 0, n, U, 3, flags)
        e[::11] = INT64_MIN
        gpu_ctx.load_columns(s, e, u, d, U)
        for now in [oracle.T0_MS - 60 * DAY, oracle.T0_MS, INT64_MIN, 2 ** 62, oracle.T0_MS - 119 * DAY, int(s.min()) + W, int(s.min()) + W - 1]:
            want = oracle.archive_queue(s, e, u, U, now, W)
            assert np.array_equal(gpu_ctx.archive_queue(now, W), want), (n, now)
    n, U = 1500000, 30011
    s, e, u, d = oracle.gen(SEED, n, 0, n, U, 32, 1)
    e[::7] = INT64_MIN
    gpu_ctx.load_columns(s, e, u, d, U)
    for now in [oracle.T0_MS - 118 * DAY, oracle.T0_MS - 119 * DAY - 20 * 3600 * 1000, oracle.T0_MS]:
        got = gpu_ctx.archive_queue(now, W)
        assert np.array_equal(got, oracle.archive_queue_numpy(s, e, u, U, now, W))
    assert got.size == np.count_nonzero(e != INT64_MIN)      # at T0 every present row belongs to an old-enough group
    # a feed scan afterwards is unaffected
    gpu_ctx.set_disciplines(ALL, 32)
    assert_same(gpu_ctx.scan(*spec_query(oracle)[:2]), oracle.scan(s, e, u, d, U, *spec_query(oracle)[:2], 0xFFFFFFFF))


def test_retention_purge_calendar_months(gpu_ctx, oracle):
    """sqlProvider.js:863-890,991-1009: now >= addMonths(createdAt, 2) with JS month arithmetic.  The device's integer
    civil-date arithmetic against (1) the JS engine's own Date results (tests/golden/addmonths_utc.json) and (2) the
    libc-based oracle on the synthetic corpus and on extreme timestamps, with and without a zone offset."""
    g = json.load(open(os.path.join(GOLDEN, "addmonths_utc.json")))
    by_months = {}
    for ts, m, want in g["cases"]:
        by_months.setdefault(m, []).append((ts, want))
    for m, lst in by_months.items():
        ts = np.array([t for t, _ in lst], np.int64)
        e = np.full(ts.size, 2 ** 62, np.int64)
        z = np.zeros(ts.size, np.int32)
        for probe in (0, -1):   # now == expiry purges (>=); one millisecond earlier does not
            for k, (t, want) in enumerate(lst):
                if want is None:
                    continue
                gpu_ctx.load_columns(ts[k:k + 1], e[k:k + 1], z[:1], z[:1], 1)
                assert gpu_ctx.retention_purge(want + probe, m).size == (1 if probe == 0 else 0), (t, m, want)
                if len(lst) > 60 and k > 40:
                    break
        # whole list at once against the oracle
        gpu_ctx.load_columns(ts, e, z, z, 1)
        now = int(np.median(ts)) + 40 * DAY
        want_rows = oracle.retention_queue(ts, e, now, m)
        assert np.array_equal(gpu_ctx.retention_purge(now, m), want_rows)
        assert gpu_ctx.retention_purge(now, m).size == 0           # purged rows are tombstoned
    n, U = 300000, 50
    s, e, u, d = oracle.gen(SEED, n, 0, n, U, 3, 1)
    e[::9] = INT64_MIN
    for tz in (0, 330 * 60000, -8 * 3600000):
        for now in (oracle.T0_MS, oracle.T0_MS - 30 * DAY, oracle.T0_MS - 200 * DAY):
            gpu_ctx.load_columns(s, e, u, d, U)
            assert np.array_equal(gpu_ctx.retention_purge(now, 2, tz), oracle.retention_queue(s, e, now, 2, tz))


def test_retention_purge_under_real_time_zones(pie, gpu_ctx, oracle):
    """`setMonth` on a LOCAL Date under daylight saving.  pie_retention_purge_tz with the transition table
    the JS host built (tests/golden/addmonths_zones.json carries it beside the vectors): (1) every JS-engine vector of every
    zone, the boundary probed to the millisecond — now = expiry purges, one millisecond earlier does not — all rows of a zone
    in ONE table per probe; (2) whole lists against the oracle's table-driven restatement; (3) a table without transitions is
    the fixed-offset entry point; (4) a table from Python's zoneinfo gives the same answers; malformed tables are refused."""
    g = json.load(open(os.path.join(GOLDEN, "addmonths_zones.json")))
    n_checked = 0
    for zone, z in g["zones"].items():
        T, off = np.array(z["transitions"], np.int64), np.array(z["offsets"], np.int64)
        by_m = {}
        for ts, m, want in z["cases"]:
            if want is not None:
                by_m.setdefault(m, []).append((ts, want))
        for m, lst in by_m.items():
            ts = np.array([t for t, _ in lst], np.int64)
            want = np.array([w for _, w in lst], np.int64)
            e = np.full(ts.size, 2 ** 62, np.int64)
            zc = np.zeros(ts.size, np.int32)
            # rows sorted by expiry: at now = want[k] exactly the rows with expiry <= want[k] go, at want[k] - 1 those below it
            order = np.argsort(want, kind="stable")
            ts, want = ts[order], want[order]
            for k in sorted(set([0, ts.size // 3, ts.size // 2, ts.size - 1])):
                for probe in (0, -1):
                    gpu_ctx.load_columns(ts, e, zc, zc, 1)
                    got = gpu_ctx.retention_purge(int(want[k]) + probe, m, tz_table=(T, off))
                    assert np.array_equal(got, np.nonzero(want <= int(want[k]) + probe)[0].astype(np.int32)), (zone, m, k, probe)
            gpu_ctx.load_columns(ts, e, zc, zc, 1)
            now = int(np.median(want))
            assert np.array_equal(gpu_ctx.retention_purge(now, m, tz_table=(T, off)), oracle.retention_queue_tz(ts, e, now, m, T, off)), (zone, m)
            n_checked += ts.size
        if T.size == 0:   # no transitions: the fixed-offset entry point gives the same rows
            ts = np.array([c[0] for c in z["cases"]], np.int64)
            e = np.full(ts.size, 2 ** 62, np.int64)
            zc = np.zeros(ts.size, np.int32)
            now = int(np.median(ts))
            gpu_ctx.load_columns(ts, e, zc, zc, 1)
            a = gpu_ctx.retention_purge(now, 2, tz_table=(T, off))
            gpu_ctx.load_columns(ts, e, zc, zc, 1)
            assert np.array_equal(a, gpu_ctx.retention_purge(now, 2, int(off[0])))
    assert n_checked > 10000
    # the synthetic corpus under New York and Lord Howe (30-minute DST), zoneinfo's table, against the oracle with the JS table
    n, U = 300000, 50
    s, e, u, d = oracle.gen(SEED, n, 0, n, U, 3, 1)
    e[::9] = INT64_MIN
    for zone in ("America/New_York", "Australia/Lord_Howe"):
        z = g["zones"][zone]
        py_table = pie.tz_table(zone, 631152000000, 2145916800000)
        for now in (oracle.T0_MS, oracle.T0_MS - 30 * DAY, oracle.T0_MS - 58 * DAY):
            gpu_ctx.load_columns(s, e, u, d, U)
            assert np.array_equal(gpu_ctx.retention_purge(now, 2, tz_table=py_table), oracle.retention_queue_tz(s, e, now, 2, z["transitions"], z["offsets"]))
    gpu_ctx.load_columns(s, e, u, d, U)
    with pytest.raises(pie.PieError):   # transitions out of order
        gpu_ctx.retention_purge(oracle.T0_MS, 2, tz_table=(np.array([5, 3], np.int64), np.array([0, 3600000, 0], np.int64)))
    with pytest.raises(ValueError):
        gpu_ctx.retention_purge(oracle.T0_MS, 2, tz_table=(np.array([5], np.int64), np.array([0], np.int64)))


def test_column_files_round_trip(pie, gpu_ctx, oracle, tmp_path):
    """Flat column files: save, load into a fresh context, same table and same feeds; bad directories fail loudly."""
    n, U = 123457, 321
    s, e, u, d = oracle.gen(SEED, n, 0, n, U, 32, 1)
    e[::13] = INT64_MIN
    gpu_ctx.load_columns(s, e, u, d, U)
    gpu_ctx.save_columns(str(tmp_path / "tbl"))
    assert sorted(os.listdir(tmp_path / "tbl")) == ["disc.i32", "end.i64", "header.json", "start.i64", "user.i32"]
    assert json.load(open(tmp_path / "tbl" / "header.json")) == {"format": "pie-columns", "version": 1, "rows": n, "users": U}
    assert np.array_equal(np.fromfile(tmp_path / "tbl" / "end.i64", np.int64), e)
    with pie.PieScan(0) as ctx2:
        ctx2.load_columns_dir(str(tmp_path / "tbl"))
        assert (ctx2.n, ctx2.n_users) == (n, U)
        for a, b in zip(ctx2.read_columns(), (s, e, u, d)):
            assert np.array_equal(a, b)
        ctx2.set_disciplines(ALL, 32)
        now, cutoff, _ = spec_query(oracle)
        assert_same(ctx2.scan(now, cutoff), oracle.scan(s, e, u, d, U, now, cutoff, 0xFFFFFFFF))
        with pytest.raises(pie.PieError):
            ctx2.load_columns_dir(str(tmp_path / "nope"))
        (tmp_path / "tbl" / "user.i32").write_bytes(b"\0" * 8)
        with pytest.raises(pie.PieError):
            ctx2.load_columns_dir(str(tmp_path / "tbl"))
    gpu_ctx.load_columns(s[:0], e[:0], u[:0], d[:0], 1)      # an empty table round-trips too
    gpu_ctx.save_columns(str(tmp_path / "empty"))
    gpu_ctx.load_columns_dir(str(tmp_path / "empty"))
    assert gpu_ctx.n == 0


@pytest.mark.parametrize("on_end", [False, True])
def test_expired_queue_parity(gpu_ctx, oracle, monkeypatch, on_end):
    """prev < end <= now -> ascending row list, on the 2-byte liveness key (rows strictly between the two keys need no
    compare, rows on either boundary key do) and, with PIE_EXPIRED_ON_END, on the `end` column itself: window edges
    at exact `end` values, windows inside one key bucket, empty and inverted windows, after touches and tombstones."""
    if on_end:
        monkeypatch.setenv("PIE_EXPIRED_ON_END", "1")
    rng = np.random.default_rng(11)
    for n, flags in [(1, 0), (257, 1), (100003, 1), (1 << 20, 0)]:
        s, e, u, d = oracle.gen(SEED, n, 0, n, 100, 32, flags)
        e = e.copy()
        gpu_ctx.load_columns(s, e, u, d, 100)
        k = int(e[n // 2])
        windows = [(oracle.T0_MS - 50 * DAY, oracle.T0_MS - 20 * DAY), (INT64_MIN, 2 ** 62), (5, 4),
                   (oracle.T0_MS - 6 * 3600 * 1000 - 60000, oracle.T0_MS - 6 * 3600 * 1000),
                   (k - 1, k), (k, k + 1), (k, k), (k - 1000, k + 1000), (INT64_MIN, k), (k, 2 ** 63 - 1)]
        for prev, now in windows:
            assert np.array_equal(gpu_ctx.expired_queue(prev, now), oracle.expired_queue(e, prev, now))
        if n > 1000:
            rows = rng.choice(n, 500, replace=False).astype(np.int32)
            new_end = rng.integers(oracle.T0_MS - 200 * DAY, oracle.T0_MS + 200 * DAY, rows.size).astype(np.int64)
            new_end[:20] = INT64_MIN
            gpu_ctx.set_end(rows, new_end)
            e[rows] = new_end
            gone = gpu_ctx.delete_user(7)
            e[gone] = INT64_MIN
            for prev, now in windows:
                assert np.array_equal(gpu_ctx.expired_queue(prev, now), oracle.expired_queue(e, prev, now))


def test_config2_parity_1e7(gpu_ctx, oracle):
    """BASELINE config 2: 10^7 sessions / 10^4 users / 32 disciplines, generated on the device."""
    n, U, D = 10 ** 7, 10 ** 4, 32
    now, cutoff, mask = spec_query(oracle)
    for flags in (0, 1, 2):
        gpu_ctx.gen_synthetic(SEED, n, 0, n, U, D, flags)
        gpu_ctx.set_disciplines(mask, D)
        got = gpu_ctx.scan(now, cutoff)
        want = oracle.scan(*oracle.gen(SEED, n, 0, n, U, D, flags), U, now, cutoff, mask & ((1 << D) - 1))
        assert_same(got, want)
        assert got[2].size > 0


def test_config4_user_hash_shards_reassemble(pie, gpu_ctx, oracle):
    """BASELINE config 4 as a parity case: the table is user-hash sharded 8 ways, every shard is scanned by the HIP
    path (one after the other on this one GPU), and the per-shard feeds reassemble to the oracle's feeds of the
    whole table — same rows, same order, for every user."""
    from sph_pie_amd.shard import partition_by_user_hash
    n, U, D, G = 2 * 10 ** 6, 5003, 32, 8
    cols = oracle.gen(SEED, n, 0, n, U, D, 1)
    now, cutoff, mask = spec_query(oracle)
    now -= 20 * DAY   # a less selective query, so most users have a feed
    wc, wo, wi = oracle.scan(*cols, U, now, cutoff, mask & 0xFFFFFFFF)
    shards = partition_by_user_hash(*cols, U, G)
    assert sum(s["rows"].size for s in shards) == n
    seen_users = 0
    for r, sh in enumerate(shards):
        gpu_ctx.load_columns(sh["start"], sh["end"], sh["user"], sh["disc"], sh["n_users"])
        gpu_ctx.set_disciplines(mask, D)
        c, o, idx = gpu_ctx.scan(now, cutoff)
        for lu, gu in enumerate(sh["users"]):
            assert pie.shard_of(int(gu), G) == r
            feed = sh["rows"][idx[o[lu]:o[lu + 1]]]
            assert np.array_equal(feed, wi[wo[gu]:wo[gu + 1]])
            seen_users += 1
    assert seen_users == U and wi.size > U


def test_full_size_properties_1e8(gpu_ctx, oracle, request):
    """BASELINE config 3 (10^8 / 10^5 / 32): properties that need no full-size oracle run, plus an exact
    oracle comparison on the selected rows only."""
    n, U, D = 10 ** 8, 10 ** 5, 32
    now, cutoff, mask = spec_query(oracle)
    gpu_ctx.gen_synthetic(SEED, n, 0, n, U, D, 0)
    gpu_ctx.set_disciplines(mask, D)
    counts, offsets, idx = gpu_ctx.scan(now, cutoff)
    m = idx.size
    assert offsets[0] == 0 and offsets[-1] == m == int(counts.sum(dtype=np.int64))
    assert np.array_equal(np.diff(offsets), counts)
    assert np.unique(idx).size == m                      # no row twice
    s, e, u, d = gpu_ctx.fetch_rows(idx)
    assert np.all(e > now) and np.all(s >= cutoff) and np.all(((mask >> d.astype(np.uint64)) & 1) == 1)
    assert np.array_equal(u, np.repeat(np.arange(U, dtype=np.int32), counts))   # buckets are per user, in user order
    key_ok = (np.diff(s) > 0) | ((np.diff(s) == 0) & (np.diff(idx) > 0)) | (np.diff(u) != 0)
    assert np.all(key_ok)                                # (start, row) ascending inside every bucket
    # selectivity of the spec query is 18 h / 120 d live x 16/32 disciplines
    assert abs(m / n - 0.5 * 18 / (120 * 24)) < 2e-4
    # idempotence: the same query again gives the same bytes
    c2, o2, i2 = gpu_ctx.scan(now, cutoff)
    assert np.array_equal(c2, counts) and np.array_equal(i2, idx)
    # exact check against the oracle over the WHOLE table (the multi-thread form of the oracle, itself checked against the
    # single-thread one in test_oracle_golden.py): counts, offsets and every row index, exact M
    cols = oracle.gen_mt(SEED, n, 0, n, U, D, 0, threads=16)
    wc, wo, wi = oracle.scan_mt(*cols, U, now, cutoff, mask & ((1 << D) - 1), 16)
    assert wi.size == m
    assert np.array_equal(counts, wc) and np.array_equal(offsets, wo) and np.array_equal(idx, wi)
    # the batched scan at full size: 16 queries, one table pass, each equal to the oracle's answer for it
    queries = [(now - 977 * q, cutoff - (q % 3) * DAY, (mask, 0xAAAAAAAAAAAAAAAA, ALL)[q % 3]) for q in range(16)]
    gpu_ctx.set_disciplines(ALL, D)
    got = gpu_ctx.scan_batch(queries)
    for q, (qn, qc, qm) in enumerate(queries):
        w = oracle.scan_mt(*cols, U, qn, qc, qm & ((1 << D) - 1), 16)
        for a, b in zip(got[q], w):
            assert np.array_equal(a, b), q
    # config 5 at full size: the expired-session dispatch queue (change predicate prev < end <= now) exact against the oracle
    # for a one-day and a one-month window, and the archive group-min chain by its defining properties + exact against numpy
    e_col = cols[1]
    for prev, nw in [(oracle.T0_MS - 30 * DAY, oracle.T0_MS - 29 * DAY), (oracle.T0_MS - 90 * DAY, oracle.T0_MS - 60 * DAY), (INT64_MIN, oracle.T0_MS - 119 * DAY)]:
        q_gpu = gpu_ctx.expired_queue(prev, nw)
        assert np.array_equal(q_gpu, oracle.expired_queue(e_col, prev, nw))
    window = 3600 * 1000
    now_a = oracle.T0_MS - 120 * DAY + 2 * window           # a user qualifies iff it has a session in the corpus' first hour: ~29 %
    q_arch = gpu_ctx.archive_queue(now_a, window)
    s_col, u_col = cols[0], cols[2]
    earliest = np.full(U, np.iinfo(np.int64).max, np.int64)
    np.minimum.at(earliest, u_col, s_col)
    qual = (now_a - earliest) >= window
    assert 0.2 < qual.mean() < 0.4
    assert q_arch.size == int(np.count_nonzero(qual[u_col]))                 # every row of every qualifying group, nothing else
    assert np.all(qual[u_col[q_arch]]) and np.unique(q_arch).size == q_arch.size
    first = np.full(U, n, np.int64)
    np.minimum.at(first, u_col, np.arange(n, dtype=np.int64))
    grp = u_col[q_arch]
    change = np.nonzero(np.diff(grp) != 0)[0] + 1
    heads = np.r_[0, change]
    assert np.unique(grp[heads]).size == heads.size                           # each group is one contiguous run
    assert np.all(np.diff(first[grp[heads]]) > 0)                             # groups in order of first appearance
    assert np.all((np.diff(q_arch) > 0) | (np.diff(grp) != 0))                # table order inside a group
    keep = np.nonzero(qual[u_col])[0]                                         # and row for row: the numpy restatement at full size
    assert np.array_equal(q_arch, keep[np.lexsort((keep, first[u_col[keep]]))].astype(np.int32))
    del keep
    # the ordered run at full size (sph-pie_amd/csrc/pie_ordered.h): the spec query (keyed form), a query that selects a
    # quarter of the table (dense form: 2.5 x 10^7 rows out) and the 16-query batch, each exact against the oracle
    gpu_ctx.set_ordered_run(2)
    request.addfinalizer(lambda: gpu_ctx.set_ordered_run(1))
    gpu_ctx.set_disciplines(mask, D)
    c3, o3, i3 = gpu_ctx.scan(now, cutoff)
    assert gpu_ctx.stats()["k1_variant"] & 0x2400 == 0x2400
    assert np.array_equal(c3, wc) and np.array_equal(o3, wo) and np.array_equal(i3, wi)
    wide_now = oracle.T0_MS - 100 * DAY
    c4, o4, i4 = gpu_ctx.scan(wide_now, cutoff)
    assert gpu_ctx.stats()["k1_variant"] == 0x2003
    w4 = oracle.scan_mt(*cols, U, wide_now, cutoff, mask & ((1 << D) - 1), 16)
    assert i4.size == w4[2].size and i4.size > n // 5
    assert np.array_equal(c4, w4[0]) and np.array_equal(o4, w4[1]) and np.array_equal(i4, w4[2])
    del w4, i4
    gpu_ctx.set_disciplines(ALL, D)
    got_run = gpu_ctx.scan_batch(queries)
    assert gpu_ctx.stats()["k1_variant"] & 0x3000 == 0x3000
    for q in range(len(queries)):
        for a, b in zip(got_run[q], got[q]):                                  # `got` was checked against the oracle above
            assert np.array_equal(a, b), q
    info = gpu_ctx.table_info()
    assert info["ordered_builds"] >= 1 and info["ordered_rows"] == n
