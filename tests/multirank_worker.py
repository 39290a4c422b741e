"""One rank of the multi-rank GPU rehearsal (tests/test_gpu_multirank.py starts `world` of these as fresh child processes,
all on GPU 0 — RCCL refuses two ranks on one device, so the transport is gloo with pinned-host staging, shard.py
transport="host"; everything else is the code a `bench.py --gpus N` run executes over RCCL: pie_shard_table on ONE corpus,
HipShardBackend, BatchedFeeds (union and lists, several steps per all-gather) and ShardedFeeds, capacity negotiation).

Every rank rebuilds the GLOBAL feeds of every query from what it gathered and compares them with the oracle's scan of the
whole, unsharded table.

usage: multirank_worker.py RANK WORLD PORT N_ROWS N_USERS OUT_DIR"""
import json
import os
import sys

import torch  # before libpie_hip.so initialises HIP (tests/conftest.py)
import torch.distributed as dist
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

T0, DAY = 1700000000000, 86400 * 1000
SEED = 0x5EED5EED


def main():
    rank, world, port, n, U, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    import oracle_py  # the checker
    import sph_pie_amd as pie
    from sph_pie_amd.shard import BatchedFeeds, HipShardBackend, ShardedFeeds, union_feed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    report = {"rank": rank, "checks": 0}
    try:
        D = 32
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        ctx = pie.PieScan(0)
        ctx.gen_synthetic(SEED, n, 0, n, U, D, 1)
        n_local, u_local = ctx.shard_table(rank, world)
        rows_g, users_g = ctx.shard_maps()
        maps = [None] * world
        dist.all_gather_object(maps, (rows_g.astype(np.int64), users_g[:u_local].astype(np.int64), u_local))
        assert sum(m[0].size for m in maps) == n
        cols = oracle_py.gen(SEED, n, 0, n, U, D, 1)
        backend = HipShardBackend(ctx, dev)
        ctx.set_disciplines(0xFFFFFFFF, D)
        queries = [(T0 - 6 * 3600 * 1000 - 977 * q, T0 - (61 + q % 2) * DAY, (0x55555555, 0xAAAAAAAA, 0xFFFFFFFF)[q % 3]) for q in range(7)]
        want = [oracle_py.scan(*cols, U, qn, qc, qm) for qn, qc, qm in queries]

        def check_lists(out, nq):
            assert out is not None
            for q in range(nq):
                wc, wo, wi = want[q]
                total = 0
                for r in range(world):
                    rows_r, users_r, u_r = maps[r]
                    off = out["offsets"][r, q].cpu().numpy()
                    m = int(out["lengths"][r, q])
                    rows = out["rows"][r, q].cpu().numpy()[:m]
                    assert off[0] == 0 and off[-1] == m
                    total += m
                    for lu in range(u_r):
                        gu = int(users_r[lu])
                        assert np.array_equal(rows_r[rows[off[lu]:off[lu + 1]]], wi[wo[gu]:wo[gu + 1]]), (rank, q, gu)
                assert total == wi.size
                report["checks"] += 1

        def check_union(out, nq):
            assert out is not None
            host = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in out.items()}
            for q in range(nq):
                wc, wo, wi = want[q]
                total = 0
                for r in range(world):
                    rows_r, users_r, u_r = maps[r]
                    for lu in range(u_r):
                        gu = int(users_r[lu])
                        rows = union_feed(host, r, q, lu).numpy()
                        total += rows.size
                        assert np.array_equal(rows_r[rows], wi[wo[gu]:wo[gu + 1]]), (rank, q, gu)
                assert total == wi.size
                report["checks"] += 1

        def steps(bf, k, qs):
            out = bf.run_steps(k, qs)
            if out is None:   # capacity raised on every rank: once more
                out = bf.run_steps(k, qs)
            return out

        for g, k in ((1, 3), (3, 7), (8, 5)):
            uf = BatchedFeeds(backend, rank, world, u_local, q_max=len(queries), union=True, steps_per_gather=g, transport="host")
            check_union(steps(uf, k, queries), len(queries))
            lf = BatchedFeeds(backend, rank, world, u_local, q_max=len(queries), union=False, steps_per_gather=g, transport="host")
            check_lists(steps(lf, k, queries), len(queries))
        # a smaller batch after a larger one on the same driver; then a query that outgrows the negotiated capacity
        uf = BatchedFeeds(backend, rank, world, u_local, q_max=len(queries), union=True, steps_per_gather=2, transport="host")
        check_union(steps(uf, 4, queries), len(queries))
        check_union(steps(uf, 3, queries[:2]), 2)
        # one query per scan: ShardedFeeds, one and several scans per all-gather
        qn, qc, qm = queries[0]
        ctx.set_disciplines(qm, D)
        wc, wo, wi = want[0]
        for batch in (1, 4):
            sf = ShardedFeeds(backend, rank, world, u_local, batch=batch, transport="host")
            out = sf.run_steps(6, qn, qc)
            if out is None:
                out = sf.run_steps(6, qn, qc)
            total = 0
            for r in range(world):
                rows_r, users_r, u_r = maps[r]
                off = out["offsets"][r].cpu().numpy()
                rows = out["rows"][r].cpu().numpy()[: int(out["lengths"][r])]
                total += rows.size
                for lu in range(u_r):
                    gu = int(users_r[lu])
                    assert np.array_equal(rows_r[rows[off[lu]:off[lu + 1]]], wi[wo[gu]:wo[gu + 1]]), (rank, gu)
            assert total == wi.size
            report["checks"] += 1
        report["ok"] = True
        report["rows_local"], report["users_local"] = n_local, u_local
        ctx.close()
    finally:
        with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
            json.dump(report, f)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
