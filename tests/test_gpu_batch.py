"""GPU: the batched scan (pie_scan_batch_*: Q queries, one table pass) against Q oracle scans on the same seeded inputs —
bit-exact counts / offsets / idx per query — including the queries a batch hands to the general path (dense queries,
buckets of more than 16 rows), mixed now / cutoff / mask, ragged sizes, two batches in flight and the per-query result
messages.  Through the C ABI (ctypes)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

INT64_MIN = -(2 ** 63)
ALL = 2 ** 64 - 1
DAY = 86400 * 1000
HOUR = 3600 * 1000
SEED = 0x5EED5EED


def assert_same(got, want, tag=""):
    for name, a, b in zip(("counts", "offsets", "idx"), got, want):
        assert a.dtype == b.dtype, (tag, name)
        assert np.array_equal(a, b), (tag, name)


def oracle_answers(oracle, cols, U, D, queries):
    s, e, u, d = cols
    lim = ALL if D >= 64 else (1 << D) - 1
    return [oracle.scan(s, e, u, d, U, now, cutoff, mask & lim) for now, cutoff, mask in queries]


def mixed_queries(oracle, k):
    """k sparse queries with distinct now / cutoff / mask (the request mix of a feed server: every request samples its own
    clock; cutoffs change daily; masks follow the caller's role)."""
    t0 = oracle.T0_MS
    masks = [0x5555555555555555, 0xAAAAAAAAAAAAAAAA, ALL, 0x00000000FFFF0000, 0x1, 0x8000000000000001]
    return [(t0 - 6 * HOUR - 977 * i - (i % 3) * HOUR, t0 - (61 + i % 4) * DAY - 13 * i, masks[i % len(masks)]) for i in range(k)]


@pytest.mark.parametrize("n,U,D,flags", [
    (1, 1, 1, 0), (65, 3, 2, 0), (4097, 9, 7, 1), (100003, 97, 32, 0), (1 << 20, 10 ** 4, 32, 0), (3000017, 20011, 64, 1),
    (1 << 20, 10 ** 4, 32, 4), (3000017, 20011, 64, 5),   # rows in order of creation: every live row at the table's end
])
@pytest.mark.parametrize("nq", [1, 3, 16])
def test_batch_equals_separate_scans(gpu_ctx, oracle, n, U, D, flags, nq):
    cols = oracle.gen(SEED, n, 0, n, U, D, flags)
    gpu_ctx.load_columns(*cols, U)
    gpu_ctx.set_disciplines(ALL, D)
    queries = mixed_queries(oracle, nq)
    got = gpu_ctx.scan_batch(queries)
    for k, (g, w) in enumerate(zip(got, oracle_answers(oracle, cols, U, D, queries))):
        assert_same(g, w, "query %d" % k)
    # the same through single scans of the product (mask per query through set_disciplines)
    for k in (0, nq - 1):
        gpu_ctx.set_disciplines(queries[k][2], D)
        assert_same(gpu_ctx.scan(queries[k][0], queries[k][1]), got[k], "single %d" % k)


def test_batch_with_queries_that_fall_back(gpu_ctx, oracle):
    """One batch holding sparse queries, a dense one (25 % of the rows selected: handed to the general path up front from the
    key histogram), an everything-selected one, a nothing-selected one and — on a table with few users — buckets of more
    than 16 rows (found by the batch's own offsets kernel, rerun on the general path).  Every answer equals the oracle's."""
    n, U, D = 1 << 20, 5000, 32
    cols = oracle.gen(SEED, n, 0, n, U, D, 0)
    gpu_ctx.load_columns(*cols, U)
    gpu_ctx.set_disciplines(ALL, D)
    t0 = oracle.T0_MS
    queries = [
        (t0 - 6 * HOUR, t0 - 61 * DAY, 0x5555555555555555),
        (t0 - 100 * DAY, t0 - 61 * DAY, 0xAAAAAAAAAAAAAAAA),   # dense
        (INT64_MIN, INT64_MIN, ALL),                            # everything
        (2 ** 62, INT64_MIN, ALL),                              # nothing
        (t0 - 5 * HOUR, INT64_MIN, ALL),
        (t0 - 7 * HOUR + 1, t0 - 30 * DAY, 0x3),
    ]
    got = gpu_ctx.scan_batch(queries)
    want = oracle_answers(oracle, cols, U, D, queries)
    for k, (g, w) in enumerate(zip(got, want)):
        assert_same(g, w, "query %d" % k)
    assert got[2][2].size == n and got[3][2].size == 0
    # few users: sparse queries whose buckets outgrow the 16 direct slots
    n, U = 300000, 7
    cols = oracle.gen(SEED, n, 0, n, U, D, 0)
    gpu_ctx.load_columns(*cols, U)
    queries = mixed_queries(oracle, 5)
    got = gpu_ctx.scan_batch(queries)
    want = oracle_answers(oracle, cols, U, D, queries)
    assert max(int(w[0].max()) for w in want) > 16
    for k, (g, w) in enumerate(zip(got, want)):
        assert_same(g, w, "query %d" % k)


def test_batch_edge_tables(gpu_ctx, oracle):
    """Tombstoned and sentinel ends, disciplines outside the table, equal `end` values (every key ambiguous), queries whose
    `now` sits exactly on an `end`, and a query below the fine key's base next to ones above it (the batch then streams the
    2-byte key)."""
    n, U, D = 200000, 300, 40
    s, e, u, d = oracle.gen(SEED, n, 0, n, U, D, 1)
    e = e.copy()
    d = d.copy()
    e[::7] = INT64_MIN
    d[::11] = 64 + (np.arange(d[::11].size) % 5)
    d[5::13] = -1
    cols = (s, e, u, d)
    gpu_ctx.load_columns(*cols, U)
    gpu_ctx.set_disciplines(ALL, 64)
    t0 = oracle.T0_MS
    live_ends = np.sort(e[e != INT64_MIN])
    top = int(live_ends[-1])
    queries = [(top - 1, INT64_MIN, ALL), (top, INT64_MIN, ALL), (int(live_ends[-50]), INT64_MIN, ALL),
               (int(live_ends[-50]) - 1, t0 - 3 * DAY, 0xF0F0F0F0F0F0F0F0), (t0 - 2 * HOUR, INT64_MIN, ALL),
               (int(live_ends[int(live_ends.size * 0.95)]), INT64_MIN, 0xFF)]
    for g, w in zip(gpu_ctx.scan_batch(queries), oracle_answers(oracle, cols, U, 64, queries)):
        assert_same(g, w)
    # all rows end at the same instant
    e2 = np.full(n, t0, np.int64)
    cols = (s, e2, u, d)
    gpu_ctx.load_columns(*cols, U)
    queries = [(t0 - 1, t0 - 119 * DAY, ALL), (t0, INT64_MIN, ALL), (t0 + 1, INT64_MIN, ALL)]
    for g, w in zip(gpu_ctx.scan_batch(queries), oracle_answers(oracle, cols, U, 64, queries)):
        assert_same(g, w)


def test_two_batches_in_flight_and_table_changes(gpu_ctx, oracle):
    """begin(i+1) before finish(i): the offsets kernels of batch i ride in the launch of batch i+1's table pass; batches of
    different sizes alternate (every span set is cleaned whatever the sizes); appends and touches between batches are seen
    by the next one."""
    n, U, D = 600011, 4099, 32
    cols = oracle.gen(SEED, n, 0, n, U, D, 0)
    gpu_ctx.load_columns(*cols, U)
    gpu_ctx.set_disciplines(ALL, D)
    sizes = [16, 2, 7, 16, 1, 9, 3]
    batches = [mixed_queries(oracle, k)[::-1] if i % 2 else mixed_queries(oracle, k) for i, k in enumerate(sizes)]
    wants = [oracle_answers(oracle, cols, U, D, qs) for qs in batches]
    gpu_ctx.scan_batch_begin(batches[0])
    for i in range(len(batches)):
        if i + 1 < len(batches):
            gpu_ctx.scan_batch_begin(batches[i + 1])
        ms = gpu_ctx.scan_batch_finish()
        assert ms == [int(w[2].size) for w in wants[i]]
        for k in range(len(batches[i])):
            assert_same(gpu_ctx.batch_read_results(k), wants[i][k], "batch %d query %d" % (i, k))
    # single scans and batches do not mix in flight
    gpu_ctx.scan_batch_begin(batches[1])
    with pytest.raises(Exception):
        gpu_ctx.scan_begin(*batches[1][0][:2])
    gpu_ctx.scan_batch_finish()
    # table changes between batches
    s, e, u, d = (c.copy() for c in cols)
    rows = np.arange(0, n, 1013, dtype=np.int32)
    new_end = np.full(rows.size, oracle.T0_MS + HOUR, np.int64)
    gpu_ctx.set_end(rows, new_end)
    e[rows] = new_end
    s2, e2, u2, d2 = oracle.gen(SEED + 1, 5000, 0, 5000, U, D, 0)
    e2 = e2 + 119 * DAY
    gpu_ctx.append_rows(s2, e2, u2, d2, U)
    cols2 = (np.concatenate([s, s2]), np.concatenate([e, e2]), np.concatenate([u, u2]), np.concatenate([d, d2]))
    qs = mixed_queries(oracle, 6)
    for g, w in zip(gpu_ctx.scan_batch(qs), oracle_answers(oracle, cols2, U, D, qs)):
        assert_same(g, w)


def test_batch_messages_equal_the_pack_kernel(pie, gpu_ctx, oracle):
    """pie_scan_batch_begin_packed: every query's result message [off[0..u_pad] | M | rows] and counts copy in caller-owned
    memory (here mapped pinned host memory) equal what the results say."""
    n, U, D = 400009, 3001, 32
    cols = oracle.gen(SEED, n, 0, n, U, D, 0)
    gpu_ctx.load_columns(*cols, U)
    gpu_ctx.set_disciplines(ALL, D)
    qs = mixed_queries(oracle, 5) + [(oracle.T0_MS - 100 * DAY, oracle.T0_MS - 61 * DAY, ALL)]   # the last one falls back
    want = oracle_answers(oracle, cols, U, D, qs)
    cap = max(int(w[2].size) for w in want) + 7
    u_pad = U + 5
    stride = u_pad + 2 + cap
    msg_h, msg_d, msg_addr = gpu_ctx.host_alloc(len(qs) * stride)
    cnt_h, cnt_d, cnt_addr = gpu_ctx.host_alloc(len(qs) * U)
    try:
        gpu_ctx.scan_batch_begin_packed(qs, msg_d, stride, u_pad, cap, cnt_d, U)
        ms, ready = gpu_ctx.scan_batch_finish(packed=True)
        assert not ready   # one query went through the general path: its message came from the pack kernel
        gpu_ctx.synchronize()
        for k, w in enumerate(want):
            m = msg_h[k * stride:(k + 1) * stride]
            assert ms[k] == w[2].size
            assert np.array_equal(m[: U + 1], w[1].astype(np.int32))
            assert np.all(m[U + 1: u_pad + 2] == w[2].size)
            assert np.array_equal(m[u_pad + 2: u_pad + 2 + w[2].size], w[2])
            assert np.array_equal(cnt_h[k * U:(k + 1) * U], w[0])
        gpu_ctx.scan_batch_begin_packed(qs[:5], msg_d, stride, u_pad, cap, cnt_d, U)
        ms, ready = gpu_ctx.scan_batch_finish(packed=True)
        assert not ready   # round 3: per-query messages are materialised from the batch's union and packed at finish
        gpu_ctx.synchronize()
        for k, w in enumerate(want[:5]):
            m = msg_h[k * stride:(k + 1) * stride]
            assert np.array_equal(m[: U + 1], w[1].astype(np.int32)) and np.array_equal(m[u_pad + 2: u_pad + 2 + w[2].size], w[2])
    finally:
        gpu_ctx.host_free(msg_addr)
        gpu_ctx.host_free(cnt_addr)


def test_batch_argument_errors(gpu_ctx, oracle):
    cols = oracle.gen(SEED, 1000, 0, 1000, 10, 3, 0)
    gpu_ctx.load_columns(*cols, 10)
    with pytest.raises(Exception):
        gpu_ctx.scan_batch_begin([])
    with pytest.raises(Exception):
        gpu_ctx.scan_batch_begin(mixed_queries(oracle, 65))
    with pytest.raises(Exception):
        gpu_ctx.scan_batch_finish()
    with pytest.raises(Exception):
        gpu_ctx.batch_read_results(99)


def test_batch_cfg2_exact(gpu_ctx, oracle):
    """BASELINE config 2 (10^7 sessions / 10^4 users / 32 disciplines): 16 queries in one pass against 16 oracle scans."""
    n, U, D = 10 ** 7, 10 ** 4, 32
    gpu_ctx.gen_synthetic(SEED, n, 0, n, U, D, 0)
    gpu_ctx.set_disciplines(ALL, D)
    cols = oracle.gen(SEED, n, 0, n, U, D, 0)
    qs = mixed_queries(oracle, 16)
    got = gpu_ctx.scan_batch(qs)
    for k, (g, w) in enumerate(zip(got, oracle_answers(oracle, cols, U, D, qs))):
        assert_same(g, w, "query %d" % k)


def test_device_sharding_equals_the_host_partition(pie, oracle):
    """pie_shard_table: the whole table sharded on the device by pie_shard_of — rows kept in table order, users re-numbered
    densely — equals the numpy partition of shard.partition_by_user_hash for every rank of worlds 1, 2, 3 and 8, maps back
    included; the shards' scans reassemble the oracle's scan of the whole table."""
    from sph_pie_amd.shard import partition_by_user_hash
    n, U, D = 700001, 2311, 32
    cols = oracle.gen(SEED, n, 0, n, U, D, 0)
    now, cutoff, mask = oracle.T0_MS - 6 * HOUR, oracle.T0_MS - 61 * DAY, 0x5555555555555555
    wc, wo, wi = oracle.scan(*cols, U, now, cutoff, mask & 0xFFFFFFFF)
    with pie.PieScan(0) as ctx:
        for world in (1, 2, 3, 8):
            shards = partition_by_user_hash(*cols, U, world)
            got_counts = np.zeros(U, np.int32)
            for rank in range(world):
                ctx.gen_synthetic(SEED, n, 0, n, U, D, 0)
                n_loc, u_loc = ctx.shard_table(rank, world)
                sh = shards[rank]
                assert n_loc == sh["rows"].size and u_loc == sh["n_users"]
                s, e, u, d = ctx.read_columns()
                assert np.array_equal(s, sh["start"]) and np.array_equal(e, sh["end"])
                assert np.array_equal(u, sh["user"]) and np.array_equal(d, sh["disc"])
                rows_g, users_g = ctx.shard_maps()
                assert np.array_equal(rows_g.astype(np.int64), sh["rows"])
                assert np.array_equal(users_g[: sh["users"].size], sh["users"])
                ctx.set_disciplines(mask, D)
                c, off, idx = ctx.scan(now, cutoff)
                for lu, gu in enumerate(sh["users"]):
                    got_counts[gu] = c[lu]
                    assert np.array_equal(rows_g[idx[off[lu]:off[lu + 1]]], wi[wo[gu]:wo[gu + 1]])
            assert np.array_equal(got_counts, wc)
        # a world with more shards than users: some shards are empty
        ctx.load_columns(cols[0][:10], cols[1][:10], np.zeros(10, np.int32), cols[3][:10], 1)
        n_loc, u_loc = ctx.shard_table(1 - pie.shard_of(0, 2), 2)
        assert n_loc == 0 and u_loc == 1
        assert ctx.scan(now, cutoff)[2].size == 0


def test_exchange_over_rccl_one_rank(pie, oracle):
    """The exchange step over a real RCCL communicator (backend "nccl", world_size 1 on this one-GPU box): ShardedFeeds
    (one query per step, also several scans per collective) and BatchedFeeds (Q queries per step) — the gathered offsets
    and row lists equal the oracle's for every query."""
    import os
    import torch
    import torch.distributed as dist
    from sph_pie_amd.shard import BatchedFeeds, HipShardBackend, ShardedFeeds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29700 + os.getpid() % 200)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n, U, D = 1 << 20, 5000, 32
        cols = oracle.gen(SEED, n, 0, n, U, D, 0)
        with pie.PieScan(0) as ctx:
            ctx.load_columns(*cols, U)
            ctx.set_disciplines(ALL, D)
            backend = HipShardBackend(ctx, dev)
            feeds = ShardedFeeds(backend, 0, 1, U, always_collective=True, batch=4)
            q1 = (oracle.T0_MS - 6 * HOUR, INT64_MIN)
            want = oracle.scan(*cols, U, q1[0], q1[1], 0xFFFFFFFF)
            for res in (feeds.scan_and_gather(*q1), feeds.run_steps(9, *q1)):
                assert int(res["lengths"][0]) == want[2].size
                assert np.array_equal(res["offsets"][0].cpu().numpy(), want[1].astype(np.int32))
                assert np.array_equal(res["rows"][0].cpu().numpy()[: want[2].size], want[2])
            queries = mixed_queries(oracle, 7)
            bf = BatchedFeeds(backend, 0, 1, U, q_max=8, always_collective=True)
            bf4 = BatchedFeeds(backend, 0, 1, U, q_max=8, always_collective=True, steps_per_gather=4)
            wants = oracle_answers(oracle, cols, U, D, queries)
            for feeds_k, k in ((bf, 1), (bf, 5), (bf4, 9), (bf4, 3)):
                res = feeds_k.run_steps(k, queries)
                assert res is not None
                for q, w in enumerate(wants):
                    m = w[2].size
                    assert int(res["lengths"][0, q]) == m
                    assert np.array_equal(res["offsets"][0, q].cpu().numpy(), w[1].astype(np.int32))
                    assert np.array_equal(res["rows"][0, q].cpu().numpy()[:m], w[2])
            # the union form of the exchange: ONE message per step (per user the union of the Q lists + a query mask per row)
            from sph_pie_amd.shard import union_feed
            near = [(oracle.T0_MS - 6 * HOUR - 977 * q, oracle.T0_MS - (61 + q % 2) * DAY, (0x5555555555555555, ALL)[q % 2]) for q in range(8)]
            wants = oracle_answers(oracle, cols, U, D, near)
            uf = BatchedFeeds(backend, 0, 1, U, q_max=8, always_collective=True, union=True)
            uf3 = BatchedFeeds(backend, 0, 1, U, q_max=8, always_collective=True, union=True, steps_per_gather=3)   # three steps per all-gather
            for feeds_k, k in ((uf, 1), (uf, 4), (uf3, 7), (uf3, 2)):
                res = feeds_k.run_steps(k, near)
                assert res is not None
                uo = res["u_offsets"][0].cpu().numpy()
                assert uo[0] == 0 and uo[U] == int(res["lengths"][0]) and np.all(np.diff(uo[: U + 1]) >= 0)
                union_len = int(res["lengths"][0])
                assert union_len < sum(w[2].size for w in wants) // 4          # that is the point: far fewer rows cross the links
                rows_all, masks_all = res["rows"][0].cpu().numpy(), res["masks"][0].cpu().numpy()
                for q, w in enumerate(wants):
                    sel = ((masks_all[:union_len] >> q) & 1) == 1
                    assert np.array_equal(rows_all[:union_len][sel], w[2]), q       # every query's whole list, in order
                    for u in (0, 17, U - 1):
                        assert np.array_equal(union_feed(res, 0, q, u).cpu().numpy(), w[2][w[1][u]:w[1][u + 1]])
    finally:
        dist.destroy_process_group()


def test_union_message_layout_capacity_and_overflow(pie, oracle):
    """pie_batch_pack_union_device: offsets padded with Mu up to u_pad, rows beyond cap dropped (Mu still says how many), a
    user whose union exceeds 32 rows makes the message say -1; the same words as a numpy restatement."""
    import torch
    n, U, D = 300000, 900, 16
    cols = oracle.gen(SEED + 5, n, 0, n, U, D, 0)
    s = cols[0]
    lim = (1 << D) - 1
    queries = [(oracle.T0_MS - 6 * HOUR - 500 * q, oracle.T0_MS - 61 * DAY, (0xFFFF, 0x0F0F, 0x00FF)[q % 3]) for q in range(5)]
    with pie.PieScan(0) as ctx:
        ctx.load_columns(*cols, U)
        ctx.set_disciplines(ALL, D)
        ctx.scan_batch(queries)
        per_user = [dict() for _ in range(U)]
        for k, (now, cutoff, mask) in enumerate(queries):
            c, off, idx = oracle.scan(*cols, U, now, cutoff, mask & lim)
            for u in range(U):
                for r in idx[off[u]:off[u + 1]]:
                    per_user[u][int(r)] = per_user[u].get(int(r), 0) | (1 << k)
        rows, masks, uoff = [], [], [0]
        for u in range(U):
            items = sorted(per_user[u].items(), key=lambda it: (int(s[it[0]]), it[0]))
            rows += [r for r, _ in items]
            masks += [m for _, m in items]
            uoff.append(len(rows))
        mu = len(rows)
        for u_pad, cap in [(U, mu), (U + 77, mu + 9), (U + 1, mu // 2)]:
            msg = torch.full((u_pad + 2 + 2 * cap,), -7, dtype=torch.int32, device="cuda:0")
            torch.cuda.synchronize()
            ctx.batch_pack_union_device(msg.data_ptr(), u_pad, cap)
            ctx.synchronize()
            a = msg.cpu().numpy()
            k = min(mu, cap)
            assert np.array_equal(a[: U + 1], np.array(uoff, np.int32)) and np.all(a[U + 1: u_pad + 2] == mu)
            assert np.array_equal(a[u_pad + 2: u_pad + 2 + k], np.array(rows[:k], np.int32)) and np.all(a[u_pad + 2 + k: u_pad + 2 + cap] == -7)
            assert np.array_equal(a[u_pad + 2 + cap: u_pad + 2 + cap + k], np.array(masks[:k], np.int32))
        # dense queries: hundreds of rows per user
        ctx.scan_batch([(oracle.T0_MS - 100 * DAY, INT64_MIN, ALL), (oracle.T0_MS - 90 * DAY, INT64_MIN, ALL)])
        msg = torch.zeros(U + 2, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        ctx.batch_pack_union_device(msg.data_ptr(), U, 0)
        ctx.synchronize()
        assert int(msg[U + 1]) == -1


def test_communicator_behind_the_c_abi(pie, oracle):
    """pie_comm_*: the sharded table + RCCL exchange through the C ABI alone (what the Node addon binds), on this box's one
    GPU: a single-process communicator of one shard and a process-per-GPU communicator of world 1 (unique id path).  The
    gathered messages (grouped ncclSend / ncclRecv, own message by local copy) equal the oracle's answers per query."""
    n, U, D = 500009, 3001, 32
    cols = oracle.gen(SEED, n, 0, n, U, D, 0)
    queries = mixed_queries(oracle, 5) + [(oracle.T0_MS - 100 * DAY, oracle.T0_MS - 61 * DAY, ALL)]
    wants = oracle_answers(oracle, cols, U, D, queries)

    def check(comm, u_pad):
        ctx = comm.ctx(0)
        assert (ctx.n, ctx.n_users) == (n, U)
        ctx.set_disciplines(ALL, D)
        for _ in range(2):
            ms = comm.scan_batch_gather(queries, u_pad)
            assert ms == [[int(w[2].size) for w in wants]]
            for q, w in enumerate(wants):
                off, idx = comm.read_gathered(0, 0, q)
                assert np.array_equal(off[: U + 1], w[1].astype(np.int32)) and np.all(off[U + 1:] == w[2].size)
                assert np.array_equal(idx, w[2])
        # the shard's context is an ordinary scan context
        ctx.set_disciplines(queries[0][2], D)
        got = ctx.scan(queries[0][0], queries[0][1])
        assert np.array_equal(got[2], wants[0][2])

    with pie.PieComm([0]) as comm:
        assert comm.world == 1
        comm.gen_synthetic_sharded(SEED, n, U, D, 0)
        check(comm, 0)
    uid = pie.PieComm.unique_id()
    assert len(uid) == 128
    with pie.PieComm.for_rank(uid, 0, 1, 0) as comm:
        comm.gen_synthetic_sharded(SEED, n, U, D, 0)
        comm.reserve(len(queries), U + 3, max(int(w[2].size) for w in wants) + 5)
        check(comm, U + 3)
    with pytest.raises(pie.PieError):
        pie.PieComm([0, 0])


def test_few_users_dense_queries_on_the_keyed_form(pie, oracle, monkeypatch):
    """Found by tools/fuzz_gpu.py (case seed 20261039): three users, one discipline, the keyed form pinned, batches whose
    queries are dense.  With rows dealt to the waves in interleaved chunks EVERY block (the last one too) can stage up
    to rows_per_block records, so the staging arrays are sized blocks x rows_per_block of the largest plan, not rows —
    an overrun there corrupted the neighbouring allocation and a later batch never published its summary."""
    monkeypatch.setenv("PIE_K1_VARIANT", "0x425")
    monkeypatch.setenv("PIE_WAIT_DEADLINE_MS", "5000")
    rng = np.random.default_rng(20261039)
    n, U, D, flags = 300000, 3, 1, 3
    cols = oracle.gen(12345, n, 0, n, U, D, flags)
    s = cols[0]
    t0 = oracle.T0_MS
    with pie.PieScan(0) as ctx:
        ctx.load_columns(*cols, U)
        ctx.set_disciplines(1, D)

        def rq():
            now = int(t0 - rng.integers(0, 20 * HOUR)) if rng.random() < 0.6 else int(t0 - rng.integers(0, 130 * DAY))
            cutoff = int(rng.choice([INT64_MIN, t0 - 61 * DAY, int(s[int(rng.integers(n))])]))
            mask = int(rng.integers(0, 2 ** 63)) if rng.random() < 0.7 else ALL
            return now, cutoff, mask
        for trial in range(8):
            batches = [[rq() for _ in range(int(rng.integers(1, 17)))] for _ in range(int(rng.integers(1, 4)))]
            ctx.scan_batch_begin(batches[0])
            for k, batch in enumerate(batches):
                if k + 1 < len(batches):
                    ctx.scan_batch_begin(batches[k + 1])
                ctx.scan_batch_finish()
                want = oracle_answers(oracle, cols, U, D, batch)
                for qi in range(len(batch)):
                    assert_same(ctx.batch_read_results(qi), want[qi], f"trial {trial} batch {k} q{qi}")


# ---------------------------------------------------------------------------------------------------------------------------
# round 3: the union is the batch's primary result; up to 64 queries per table pass

def union_from_oracle(oracle, cols, U, D, queries):
    """numpy restatement of the union result: per user the rows any query selects, in (start, row) order, with a query mask per
    row.  -> (uoff[U+1], rows, masks uint64)"""
    s = cols[0]
    sel = {}
    for q, (c, off, idx) in enumerate(oracle_answers(oracle, cols, U, D, queries)):
        for r in idx:
            sel[int(r)] = sel.get(int(r), 0) | (1 << q)
    rows = np.array(sorted(sel, key=lambda r: (int(cols[2][r]), int(s[r]), r)), np.int64)
    users = cols[2][rows] if rows.size else np.zeros(0, np.int32)
    uoff = np.zeros(U + 1, np.int64)
    np.add.at(uoff, users.astype(np.int64) + 1, 1)
    return np.cumsum(uoff), rows.astype(np.int32), np.array([sel[int(r)] for r in rows], np.uint64)


@pytest.mark.parametrize("n,U,D,flags", [(65, 3, 2, 0), (100003, 97, 32, 0), (1 << 20, 10 ** 4, 32, 0), (3000017, 20011, 64, 5)])
@pytest.mark.parametrize("nq", [1, 16, 33, 64])
def test_union_is_the_primary_result(gpu_ctx, oracle, n, U, D, flags, nq):
    """pie_batch_read_union against the numpy restatement word for word; every query's M; a user's feed read from the union;
    the per-query lists materialised from it (counts / offsets / idx) against the oracle — for 1 .. 64 queries per pass."""
    cols = oracle.gen(SEED, n, 0, n, U, D, flags)
    gpu_ctx.load_columns(*cols, U)
    gpu_ctx.set_disciplines(ALL, D)
    queries = mixed_queries(oracle, nq)
    gpu_ctx.scan_batch_begin(queries)
    ms = gpu_ctx.scan_batch_finish()
    want = oracle_answers(oracle, cols, U, D, queries)
    un = gpu_ctx.batch_read_union()
    if max(int(w[0].max()) if w[0].size else 0 for w in want) <= 16 and un is not None:
        w_uoff, w_rows, w_masks = union_from_oracle(oracle, cols, U, D, queries)
        assert np.array_equal(un[0], w_uoff) and np.array_equal(un[1], w_rows) and np.array_equal(un[2], w_masks)
    assert ms == [int(w[2].size) for w in want]
    rng = np.random.default_rng(nq)
    for u in [0, U - 1] + [int(x) for x in rng.integers(0, U, 6)]:     # per-request reads, straight from the union
        for q in {0, nq - 1, nq // 2}:
            c, off, idx = want[q]
            assert np.array_equal(gpu_ctx.batch_read_user_feed(q, u), idx[off[u]:off[u + 1]]), (q, u)
    for q in sorted({0, nq - 1, nq // 2, min(nq - 1, 32), min(nq - 1, 31)}):   # materialised on request, in any order
        assert_same(gpu_ctx.batch_read_results(q), want[q], "query %d" % q)


def test_batch_of_only_dense_queries_makes_no_table_pass(gpu_ctx, oracle):
    """every query of the batch is dense (handed to the general path from the key histogram): the batched pass
    must not run at all — no candidate rows — and the answers still equal the oracle's."""
    n, U, D = 1 << 20, 5000, 32
    cols = oracle.gen(SEED, n, 0, n, U, D, 0)
    gpu_ctx.load_columns(*cols, U)
    gpu_ctx.set_disciplines(ALL, D)
    t0 = oracle.T0_MS
    queries = [(t0 - 100 * DAY, t0 - 61 * DAY, 0xAAAAAAAAAAAAAAAA), (INT64_MIN, INT64_MIN, ALL), (t0 - 90 * DAY, INT64_MIN, 0xFF)]
    got = gpu_ctx.scan_batch(queries)
    for g, w in zip(got, oracle_answers(oracle, cols, U, D, queries)):
        assert_same(g, w)
    st = gpu_ctx.stats()
    assert st["candidates"] == 0 and st["k1_variant"] == 0      # no batched pass ran
    assert gpu_ctx.batch_read_union() is None                   # per-query results only


def test_union_message_written_by_the_batch(gpu_ctx, oracle, pie):
    """pie_scan_batch_begin_union: the tail kernel writes [uoff | Mu | rows | mask_lo | mask_hi] itself (ready = 1), for 16 and
    for 64 queries, two batches in flight, into mapped host memory; equal to the union read back and to the oracle."""
    n, U, D = 500009, 3001, 32
    cols = oracle.gen(SEED, n, 0, n, U, D, 1)
    gpu_ctx.load_columns(*cols, U)
    gpu_ctx.set_disciplines(ALL, D)
    for nq in (16, 64):
        queries = mixed_queries(oracle, nq)
        w_uoff, w_rows, w_masks = union_from_oracle(oracle, cols, U, D, queries)
        mu = int(w_rows.size)
        u_pad, cap = U + 5, mu + 7
        words = u_pad + 2 + (3 if nq > 32 else 2) * cap
        bufs = [gpu_ctx.host_alloc(words) for _ in range(2)]
        for h, _, _ in bufs:
            h[:] = -7
        gpu_ctx.scan_batch_begin_union(queries, bufs[0][1], u_pad, cap)
        gpu_ctx.scan_batch_begin_union(queries, bufs[1][1], u_pad, cap)
        for h, _, _ in bufs:
            ms, ready = gpu_ctx.scan_batch_finish(packed=True)
            assert ready
            assert np.array_equal(h[: U + 1], w_uoff.astype(np.int32)) and np.all(h[U + 1: u_pad + 2] == mu)
            assert np.array_equal(h[u_pad + 2: u_pad + 2 + mu], w_rows)
            lo = h[u_pad + 2 + cap: u_pad + 2 + cap + mu].astype(np.uint32).astype(np.uint64)
            hi = h[u_pad + 2 + 2 * cap: u_pad + 2 + 2 * cap + mu].astype(np.uint32).astype(np.uint64) if nq > 32 else 0
            assert np.array_equal(lo | (hi << np.uint64(32)) if nq > 32 else lo, w_masks)
        # a message too small for the rows: truncated, Mu still says how many there are
        small = gpu_ctx.host_alloc(u_pad + 2 + 3 * 10)
        small[0][:] = -7
        gpu_ctx.scan_batch_begin_union(queries, small[1], u_pad, 10)
        ms, ready = gpu_ctx.scan_batch_finish(packed=True)
        assert ready and int(small[0][u_pad + 1]) == mu and np.array_equal(small[0][u_pad + 2: u_pad + 12], w_rows[:10])
        for _, _, addr in bufs + [small]:
            gpu_ctx.host_free(addr)


def test_requests_of_a_batch_fetched_together(gpu_ctx, oracle):
    """pie_batch_fetch_requests: many (query, user) requests of one batch in ONE call — every request's rows in feed order and
    their start / end / disc columns — against the oracle; also for a batch without a union (a query fell back) and for users
    outside the table."""
    n, U, D = 400009, 3001, 32
    cols = oracle.gen(SEED, n, 0, n, U, D, 1)
    s, e, u, d = cols
    gpu_ctx.load_columns(*cols, U)
    gpu_ctx.set_disciplines(ALL, D)
    rng = np.random.default_rng(7)
    for queries in (mixed_queries(oracle, 40), mixed_queries(oracle, 5) + [(oracle.T0_MS - 100 * DAY, oracle.T0_MS - 61 * DAY, ALL)]):
        want = oracle_answers(oracle, cols, U, D, queries)
        gpu_ctx.scan_batch_begin(queries)
        gpu_ctx.scan_batch_finish()
        nreq = 500
        qis = rng.integers(0, len(queries), nreq).astype(np.int32)
        users = rng.integers(-2, U + 3, nreq).astype(np.int32)
        off, idx, st, en, di = gpu_ctx.batch_fetch_requests(qis, users)
        assert off[0] == 0 and off[-1] == idx.size
        for i in range(nreq):
            uu, q = int(users[i]), int(qis[i])
            rows = idx[off[i]:off[i + 1]]
            if uu < 0 or uu >= U:
                assert rows.size == 0
                continue
            c, o, ix = want[q]
            assert np.array_equal(rows, ix[o[uu]:o[uu + 1]]), (i, q, uu)
        assert np.array_equal(st, s[idx]) and np.array_equal(en, e[idx]) and np.array_equal(di, d[idx])
    # too small a buffer: refused with the size it needs
    with pytest.raises(Exception):
        gpu_ctx.batch_fetch_requests(qis, users, cap_rows=3)


def test_three_batches_in_flight(gpu_ctx, oracle, pie):
    """Up to three batches may be begun before the first is finished (the next launch is queued before the host waits for a
    summary); a fourth is refused; every batch's results are exact and stay readable until three more have begun."""
    n, U, D = 700001, 4001, 32
    cols = oracle.gen(SEED, n, 0, n, U, D, 0)
    gpu_ctx.load_columns(*cols, U)
    gpu_ctx.set_disciplines(ALL, D)
    gpu_ctx.set_batch_lanes(1)   # one lane: three slots (test_batch_lanes covers more)
    sets = [mixed_queries(oracle, k) for k in (5, 40, 16, 64, 3, 33)]
    wants = [oracle_answers(oracle, cols, U, D, qs) for qs in sets]
    begun = done = 0
    while done < len(sets):
        while begun < len(sets) and begun - done < 3:
            gpu_ctx.scan_batch_begin(sets[begun])
            begun += 1
        if begun - done == 3:
            with pytest.raises(pie.PieError) as ei:
                gpu_ctx.scan_batch_begin(sets[0])
            assert ei.value.code == -6
        ms = gpu_ctx.scan_batch_finish()
        assert ms == [int(w[2].size) for w in wants[done]]
        for q in (0, len(sets[done]) - 1, len(sets[done]) // 2):
            assert_same(gpu_ctx.batch_read_results(q), wants[done][q], "batch %d query %d" % (done, q))
        un = gpu_ctx.batch_read_union()
        assert un is not None and un[1].size >= max(ms)
        done += 1
    gpu_ctx.set_batch_lanes(0)


@pytest.mark.parametrize("lanes", [2, 3, 4])
def test_batch_lanes(gpu_ctx, oracle, pie, lanes):
    """Batch lanes (pie_set_batch_lanes): batches are dealt to independent streams and run side by side; up to three per lane are
    in flight, one more is refused, finish returns them in the order they were begun, and every result — per-query lists, the
    union, a user's feed, the union message written by the batch's own tail into mapped host memory — is exact whatever lane
    the batch ran on.  The mix holds a dense query (rerun on the general path, on the main stream, while other lanes' batches
    fly) and changes the lane count with batches in flight."""
    n, U, D = 900001, 5003, 32
    cols = oracle.gen(SEED, n, 0, n, U, D, 0)
    gpu_ctx.load_columns(*cols, U)
    gpu_ctx.set_disciplines(ALL, D)
    gpu_ctx.set_batch_lanes(lanes)
    assert gpu_ctx.batch_lanes() == lanes
    t0 = oracle.T0_MS
    sizes = (5, 40, 16, 64, 3, 33, 64, 1, 17, 48, 2, 64, 9, 31)
    sets = [mixed_queries(oracle, k) for k in sizes]
    sets[4] = sets[4] + [(t0 - 100 * DAY, t0 - 61 * DAY, 0xAAAAAAAAAAAAAAAA)]   # a dense query: falls back inside finish
    wants = [oracle_answers(oracle, cols, U, D, qs) for qs in sets]
    L = U + 2 + 3 * 60000
    host, dev, host_addr = gpu_ctx.host_alloc(L * 3 * lanes)
    msgs = host.reshape(3 * lanes, L)
    try:
        begun = done = 0
        cap = 3 * lanes
        while done < len(sets):
            while begun < len(sets) and begun - done < cap:
                if begun % 2:
                    gpu_ctx.scan_batch_begin_union(sets[begun], dev + 4 * L * (begun % (3 * lanes)), U, 60000)
                else:
                    gpu_ctx.scan_batch_begin(sets[begun])
                begun += 1
            if begun - done == cap and done < 5:   # every lane holds three: one more is refused
                with pytest.raises(pie.PieError) as ei:
                    gpu_ctx.scan_batch_begin(sets[0])
                assert ei.value.code == -6
            if begun == len(sets) and done == begun - gpu_ctx.batch_lanes():
                gpu_ctx.scan_batch_flush()   # the burst is over: the lanes' waiting tails are queued together (results unchanged)
            if done == 3:
                gpu_ctx.scan_batch_flush()   # ... and a flush in mid-stream only means the next begins carry no tail
            ms, ready = gpu_ctx.scan_batch_finish(packed=True)
            want = wants[done]
            assert ms == [int(w[2].size) for w in want], done
            nq = len(sets[done])
            for q in (0, nq - 1, nq // 2):
                assert_same(gpu_ctx.batch_read_results(q), want[q], "batch %d query %d" % (done, q))
            uu = (done * 977) % U
            c, o, ix = want[nq - 1]
            assert np.array_equal(gpu_ctx.batch_read_user_feed(nq - 1, uu), ix[o[uu]:o[uu + 1]])
            un = gpu_ctx.batch_read_union()
            if done % 2 and un is not None:   # the message the tail wrote (or, not ready, the one packed behind it)
                if not ready:
                    gpu_ctx.synchronize()
                m = msgs[done % (3 * lanes)]
                mu = int(m[U + 1])
                assert mu == un[1].size
                assert np.array_equal(m[: U + 1], un[0].astype(np.int32))
                assert np.array_equal(m[U + 2: U + 2 + mu], un[1])
                lo = m[U + 2 + 60000: U + 2 + 60000 + mu].view(np.uint32).astype(np.uint64)
                if nq > 32:
                    lo |= m[U + 2 + 120000: U + 2 + 120000 + mu].view(np.uint32).astype(np.uint64) << np.uint64(32)
                assert np.array_equal(lo, un[2])
            done += 1
            if done == 5:
                gpu_ctx.set_batch_lanes(1)       # batches in flight stay on their lanes
            if done == 8:
                gpu_ctx.set_batch_lanes(lanes)
            cap = 3 * gpu_ctx.batch_lanes() if done >= 8 or done < 5 else 3
    finally:
        gpu_ctx.synchronize()
        gpu_ctx.host_free(host_addr)
        gpu_ctx.set_batch_lanes(0)


def test_flush_keeps_tails_in_batch_order(gpu_ctx, oracle):
    """pie_scan_batch_flush behind a batch that only falls back.  Such a batch (every query dense) queues no launch, so the batch
    before it keeps its tail until its finish; a flush must queue that tail BEFORE the younger batch's (a tail zeroes the span
    of the batch after the next — here the first batch's own histogram).  Found by the differential fuzz; one lane and four."""
    n, U, D = 300007, 2003, 32
    cols = oracle.gen(SEED, n, 0, n, U, D, 0)
    gpu_ctx.load_columns(*cols, U)
    gpu_ctx.set_disciplines(ALL, D)
    t0 = oracle.T0_MS
    a, c = mixed_queries(oracle, 40), mixed_queries(oracle, 7)
    dense = [(t0 - 100 * DAY, INT64_MIN, ALL), (t0 - 90 * DAY, t0 - 110 * DAY, 0xFFFF)]
    want = {id(q): oracle_answers(oracle, cols, U, D, q) for q in (a, dense, c)}
    try:
        for lanes in (1, 4):
            gpu_ctx.set_batch_lanes(lanes)
            for order in ((a, dense, c), (a, dense, dense), (dense, a, c)):
                if gpu_ctx.batch_room() < 3:
                    continue
                for q in order:
                    gpu_ctx.scan_batch_begin(q)
                gpu_ctx.scan_batch_flush()
                for q in order:
                    ms = gpu_ctx.scan_batch_finish()
                    assert ms == [int(w[2].size) for w in want[id(q)]], (lanes, len(q))
                    for qi in (0, len(q) - 1):
                        assert_same(gpu_ctx.batch_read_results(qi), want[id(q)][qi], "lanes %d query %d" % (lanes, qi))
    finally:
        gpu_ctx.set_batch_lanes(0)
