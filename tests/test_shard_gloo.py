"""CPU, world_size 2, gloo: the multi-GPU path (user-hash partition -> per-rank scan -> all-gather of counts and
row lists -> global feeds).  The per-rank scan is injected here (the CPU oracle stands in for the GPU, as the
checker the tests are allowed to use); on a GPU box the same ShardedFeeds code runs over HipShardBackend."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO

T0, DAY = 1700000000000, 86400 * 1000
INT64_MIN = -(2 ** 63)


class OracleBackend:
    """Test stand-in for HipShardBackend: same contract, CPU tensors."""
    device = "cpu"

    def __init__(self, oracle, shard, mask):
        self.o, self.sh, self.mask = oracle, shard, mask

    def scan_begin(self, now, cutoff):
        self.q = (now, cutoff)

    def scan_finish_packed(self, dst, u_pad, cap):
        now, cutoff = self.q
        c, off, idx = self.o.scan(self.sh["start"], self.sh["end"], self.sh["user"], self.sh["disc"], self.sh["n_users"],
                                  now, cutoff, self.mask)
        m = idx.size
        k = min(m, cap)
        dst[: off.size] = torch.from_numpy(off.astype(np.int32))      # off[0..U]
        dst[off.size:u_pad + 1] = m                                    # padding users: empty feeds at the end
        dst[u_pad + 1] = m
        dst[u_pad + 2:u_pad + 2 + k] = torch.from_numpy(idx[:k])
        return m


class DirectOracleBackend(OracleBackend):
    """Stand-in for the GPU backend's second contract: the scan writes its message into the buffer handed to scan_begin
    and scan_finish_packed returns (M, ready)."""
    direct_message = True

    def scan_begin(self, now, cutoff, dst=None, u_pad=0, cap=0):
        self.pending = getattr(self, "pending", [])
        self.pending.append((now, cutoff, dst, u_pad, cap))

    def scan_finish_packed(self, dst, u_pad, cap):
        now, cutoff, own_dst, own_pad, own_cap = self.pending.pop(0)
        self.q = (now, cutoff)
        if own_dst is None:                       # begun without a buffer (the capacity probe): pack into the one given
            return OracleBackend.scan_finish_packed(self, dst, u_pad, cap), False
        assert own_dst.data_ptr() == dst.data_ptr() and own_pad == u_pad and own_cap == cap
        return OracleBackend.scan_finish_packed(self, own_dst, own_pad, own_cap), True


class OracleBatchBackend:
    """Test stand-in for HipShardBackend's batched contract (batch_begin / batch_finish), CPU tensors: Q messages back to
    back in the buffer handed to batch_begin, each [off[0..u_pad] | M | rows[0..cap)]."""
    device = "cpu"

    def __init__(self, oracle, shard, n_disc):
        self.o, self.sh, self.lim = oracle, shard, (1 << n_disc) - 1
        self.pending = []

    def batch_begin(self, queries, dst=None, stride=0, u_pad=0, cap=0):
        self.pending.append((list(queries), dst, stride, u_pad, cap))

    def batch_finish(self):
        queries, dst, stride, u_pad, cap = self.pending.pop(0)
        ms = []
        for k, (now, cutoff, mask) in enumerate(queries):
            c, off, idx = self.o.scan(self.sh["start"], self.sh["end"], self.sh["user"], self.sh["disc"], self.sh["n_users"],
                                      now, cutoff, mask & self.lim)
            ms.append(int(idx.size))
            if dst is not None:
                m = dst[k * stride:(k + 1) * stride]
                kk = min(idx.size, cap)
                m[: off.size] = torch.from_numpy(off.astype(np.int32))
                m[off.size:u_pad + 2] = int(idx.size)
                m[u_pad + 2:u_pad + 2 + kk] = torch.from_numpy(idx[:kk])
        self.last = queries
        return ms, True

    def batch_pack_union(self, dst, u_pad, cap):
        """numpy restatement of pie_batch_pack_union_device over the last finished batch"""
        sh, U = self.sh, self.sh["n_users"]
        per_user = [dict() for _ in range(U)]
        for k, (now, cutoff, mask) in enumerate(self.last):
            c, off, idx = self.o.scan(sh["start"], sh["end"], sh["user"], sh["disc"], U, now, cutoff, mask & self.lim)
            for u in range(U):
                for r in idx[off[u]:off[u + 1]]:
                    per_user[u][int(r)] = per_user[u].get(int(r), 0) | (1 << k)
        rows, masks, uoff = [], [], [0]
        over = False
        for u in range(U):
            items = sorted(per_user[u].items(), key=lambda it: (int(sh["start"][it[0]]), it[0]))
            over = over or len(items) > 32
            rows += [r for r, _ in items]
            masks += [m for _, m in items]
            uoff.append(len(rows))
        mu = len(rows)
        dst[: U + 1] = torch.tensor(uoff, dtype=torch.int32)
        dst[U + 1:u_pad + 1] = mu
        dst[u_pad + 1] = -1 if over else mu
        kk = min(mu, cap)
        if kk:
            dst[u_pad + 2:u_pad + 2 + kk] = torch.tensor(rows[:kk], dtype=torch.int32)
            dst[u_pad + 2 + cap:u_pad + 2 + cap + kk] = torch.tensor(masks[:kk], dtype=torch.int32)


def _batch_worker(rank, world, port, tmp, n, U):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_py
    import sph_pie_amd  # noqa: F401
    from sph_pie_amd.shard import BatchedFeeds, partition_by_user_hash
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D = 32
        cols = oracle_py.gen(0x5EED5EED, n, 0, n, U, D, 1)
        shards = partition_by_user_hash(*cols, U, world)
        sh = shards[rank]
        feeds = BatchedFeeds(OracleBatchBackend(oracle_py, sh, D), rank, world, sh["n_users"], q_max=6)
        sparse = [(T0 - 6 * 3600 * 1000 - 977 * q, T0 - (61 + q % 2) * DAY, (0x55555555, 0xAAAAAAAA, 0xFFFFFFFF)[q % 3]) for q in range(5)]
        dense = [(INT64_MIN, INT64_MIN, 0xFFFFFFFF), (T0 - 100 * DAY, T0 - 61 * DAY, 0x0F0F0F0F)]

        def check(out, queries):
            assert out is not None
            for q, (now, cutoff, mask) in enumerate(queries):
                wc, wo, wi = oracle_py.scan(*cols, U, now, cutoff, mask)
                got_counts = np.zeros(U, np.int32)
                total = 0
                for r in range(world):
                    off = out["offsets"][r, q].numpy()
                    m = int(out["lengths"][r, q])
                    rows = out["rows"][r, q].numpy()[:m]
                    assert off[0] == 0 and off[-1] == m and np.all(np.diff(off) >= 0)
                    total += m
                    for lu, gu in enumerate(shards[r]["users"]):
                        got_counts[gu] = off[lu + 1] - off[lu]
                        assert np.array_equal(shards[r]["rows"][rows[off[lu]:off[lu + 1]]], wi[wo[gu]:wo[gu + 1]]), (rank, q, gu)
                assert np.array_equal(got_counts, wc) and total == wi.size

        check(feeds.run_steps(1, sparse), sparse)
        check(feeds.run_steps(6, sparse), sparse)
        check(feeds.run_steps(3, sparse[:2]), sparse[:2])
        assert feeds.run_steps(2, dense) is None     # outgrew the negotiated capacity on every rank: raised, call again
        check(feeds.run_steps(2, dense), dense)
        check(feeds.run_steps(4, sparse), sparse)
        # the union form of the exchange: one message per step, every query's feed a filter of it
        from sph_pie_amd.shard import UnionOverflow, union_feed
        ufeeds = BatchedFeeds(OracleBatchBackend(oracle_py, sh, D), rank, world, sh["n_users"], q_max=6, union=True)

        def check_union(out, queries):
            assert out is not None
            for q, (now, cutoff, mask) in enumerate(queries):
                wc, wo, wi = oracle_py.scan(*cols, U, now, cutoff, mask)
                total = 0
                for r in range(world):
                    assert int(out["u_offsets"][r, 0]) == 0 and int(out["u_offsets"][r, -1]) == int(out["lengths"][r])
                    for lu, gu in enumerate(shards[r]["users"]):
                        rows = union_feed(out, r, q, lu).numpy()
                        total += rows.size
                        assert np.array_equal(shards[r]["rows"][rows], wi[wo[gu]:wo[gu + 1]]), (rank, q, gu)
                assert total == wi.size

        check_union(ufeeds.run_steps(1, sparse), sparse)
        check_union(ufeeds.run_steps(5, sparse), sparse)
        # several steps per all-gather (the collective's fixed cost is shared), for group sizes that do and do not divide k
        for g, k in ((3, 7), (4, 4), (2, 1), (8, 5)):
            gf = BatchedFeeds(OracleBatchBackend(oracle_py, sh, D), rank, world, sh["n_users"], q_max=6, union=True, steps_per_gather=g)
            check_union(gf.run_steps(k, sparse), sparse)
            gl = BatchedFeeds(OracleBatchBackend(oracle_py, sh, D), rank, world, sh["n_users"], q_max=6, steps_per_gather=g)
            check(gl.run_steps(k, sparse), sparse)
            assert gl.run_steps(g + 1, dense) is None   # a group that outgrows the capacity: raised for all of it, call again
            check(gl.run_steps(g + 1, dense), dense)
        # a backend with batch lanes (the HIP library on a shard-sized table): twelve batches in flight, bounded by two message
        # groups — the driver asks batch_room before every begin and never exceeds what the backend takes
        class DeepBackend(OracleBatchBackend):
            max_seen = 0

            def batch_depth(self):
                return 12

            def batch_room(self):
                return 12 - len(self.pending)

            def batch_begin(self, queries, dst=None, stride=0, u_pad=0, cap=0):
                assert len(self.pending) < 12
                OracleBatchBackend.batch_begin(self, queries, dst, stride, u_pad, cap)
                DeepBackend.max_seen = max(DeepBackend.max_seen, len(self.pending))

        for g, k in ((8, 30), (3, 11), (1, 5)):
            DeepBackend.max_seen = 0
            df = BatchedFeeds(DeepBackend(oracle_py, sh, D), rank, world, sh["n_users"], q_max=6, union=True, steps_per_gather=g)
            check_union(df.run_steps(k, sparse), sparse)
            assert DeepBackend.max_seen == min(12, 2 * g, k) if g > 1 else DeepBackend.max_seen == 2, (g, k, DeepBackend.max_seen)
            dl = BatchedFeeds(DeepBackend(oracle_py, sh, D), rank, world, sh["n_users"], q_max=6, steps_per_gather=g)
            check(dl.run_steps(k, sparse), sparse)
        check_union(ufeeds.run_steps(3, sparse[:2]), sparse[:2])
        try:   # dense queries: hundreds of rows per user — the union form declines, the caller uses the lists
            ufeeds.run_steps(1, dense)
            raised = False
        except UnionOverflow:
            raised = True
        assert raised
        open(os.path.join(tmp, "bok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,U,world", [(20000, 37, 2), (9000, 11, 3)])
def test_batched_feeds_gloo(tmp_path, oracle, n, U, world):
    """The batched exchange driver (Q queries per step, one all-gather of the Q messages) over gloo: global feeds of every
    query rebuilt from the gathered buffers equal the oracle's on the whole table; capacity overflow path included."""
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_batch_worker, args=(world, port, str(tmp_path), n, U), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("bok%d" % r)) for r in range(world))


def _worker(rank, world, port, tmp, n, U):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_py
    import sph_pie_amd  # noqa: F401
    from sph_pie_amd.shard import ShardedFeeds, partition_by_user_hash
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cols = oracle_py.gen(0x5EED5EED, n, 0, n, U, 32, 1)
        shards = partition_by_user_hash(*cols, U, world)
        sh = shards[rank]
        mask = 0x55555555
        backend_cls = DirectOracleBackend if os.environ.get("PIE_TEST_DIRECT_BACKEND") == "1" else OracleBackend
        feeds = ShardedFeeds(backend_cls(oracle_py, sh, mask), rank, world, sh["n_users"])
        queries = [(T0 - 6 * 3600 * 1000, T0 - 61 * DAY), (INT64_MIN, INT64_MIN), (T0 - 100 * DAY, T0 - 61 * DAY)]
        # pipelined use: submit step i+1 before collecting step i (the second query outgrows the negotiated
        # capacity on purpose, so the resubmit path runs too)
        tickets = [feeds.submit(*queries[0]), feeds.submit(*queries[0])]
        first = [feeds.collect(t) for t in tickets]
        assert first[0] is not None and torch.equal(first[0]["rows"], first[1]["rows"])
        piped = feeds.run_steps(5, *queries[0])
        assert torch.equal(piped["rows"], first[0]["rows"]) and torch.equal(piped["offsets"], first[0]["offsets"])
        # several scans per all-gather (direct-message backends): same lists, for batch sizes that do and do not divide k
        for batch, k in ((3, 7), (4, 4), (2, 1)):
            feeds.batch = batch
            got = feeds.run_steps(k, *queries[0])
            assert torch.equal(got["rows"][:, : first[0]["rows"].shape[1]], first[0]["rows"]) and torch.equal(got["offsets"], first[0]["offsets"])
            assert torch.equal(got["lengths"], first[0]["lengths"])
        feeds.batch = 1
        for now, cutoff in queries:
            out = feeds.scan_and_gather(now, cutoff)
            # rebuild global feeds from the gathered buffers and compare with the oracle on the WHOLE table
            wc, wo, wi = oracle_py.scan(*cols, U, now, cutoff, mask)
            got_counts = np.zeros(U, np.int32)
            got_feeds = {}
            for r in range(world):
                off = out["offsets"][r].numpy()
                rows = out["rows"][r].numpy()[: int(out["lengths"][r])]
                assert off[0] == 0 and off[-1] == int(out["lengths"][r]) and np.all(np.diff(off) >= 0)
                for lu, gu in enumerate(shards[r]["users"]):
                    got_counts[gu] = off[lu + 1] - off[lu]
                    got_feeds[int(gu)] = shards[r]["rows"][rows[off[lu]:off[lu + 1]]]
                assert np.all(off[len(shards[r]["users"]):] == off[-1])   # padding users stay empty
            assert np.array_equal(got_counts, wc)
            for gu in range(U):
                assert np.array_equal(got_feeds.get(gu, np.zeros(0, np.int64)), wi[wo[gu]:wo[gu + 1]]), (rank, gu)
            assert int(out["lengths"].sum()) == wi.size
        # multi-rank expired-session dispatch queue: per-shard ordered queues -> one global ascending order
        from sph_pie_amd.shard import gather_expired_queues
        for prev, now in [(T0 - 50 * DAY, T0 - 20 * DAY), (INT64_MIN, 2 ** 62), (5, 4)]:
            local = oracle_py.expired_queue(sh["end"], prev, now)
            merged = gather_expired_queues(local, sh["rows"], rank, world)
            assert np.array_equal(merged, oracle_py.expired_queue(cols[1], prev, now).astype(np.int64))
        # multi-rank archive queue (the reference's group-min chain): groups = users live whole on one rank; the global
        # queue orders qualifying groups by first appearance in the whole table
        from sph_pie_amd.shard import gather_archive_queues
        for now, window in [(T0 - 30 * DAY, 43200000), (T0, 100 * DAY), (T0 - 200 * DAY, 0), (2 ** 62, 1)]:
            local = oracle_py.archive_queue(sh["start"], sh["end"], sh["user"], sh["n_users"], now, window)
            merged = gather_archive_queues(local, sh["user"], sh["rows"], rank, world)
            want = oracle_py.archive_queue(cols[0], cols[1], cols[2], U, now, window).astype(np.int64)
            assert np.array_equal(merged, want), (now, window, merged[:10], want[:10])
        open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,U,direct,world", [(20000, 37, False, 2), (3000, 5, False, 2), (20000, 37, True, 2), (12000, 101, True, 5)])
def test_sharded_feeds_world2_gloo(tmp_path, oracle, monkeypatch, n, U, direct, world):
    """world 2 (and 5) over gloo; `direct` = the backend contract of the GPU path (the scan writes the message it is handed
    at scan_begin, two scans queued ahead of the gathers, rotating buffer sets, several scans per all-gather)."""
    monkeypatch.setenv("PIE_TEST_DIRECT_BACKEND", "1" if direct else "0")
    port = 29500 + (os.getpid() % 2000) + (n % 7) + (11 if direct else 0) + 3 * world
    mp.spawn(_worker, args=(world, port, str(tmp_path), n, U), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))


def test_partition_is_a_partition(pie, oracle):
    from sph_pie_amd.shard import partition_by_user_hash
    n, U = 5000, 41
    cols = oracle.gen(0x5EED5EED, n, 0, n, U, 7, 0)
    for world in (1, 2, 4, 8):
        shards = partition_by_user_hash(*cols, U, world)
        rows = np.concatenate([s["rows"] for s in shards])
        assert np.array_equal(np.sort(rows), np.arange(n))
        users = np.concatenate([s["users"] for s in shards])
        assert np.array_equal(np.sort(users), np.arange(U))
        for r, s in enumerate(shards):
            assert all(pie.shard_of(int(g), world) == r for g in s["users"])
            assert np.array_equal(s["users"][s["user"]], cols[2][s["rows"]])  # local -> global user round trip
