"""The C-ABI communicator (pie_comm_*) with world > 1 on a one-GPU box: a fresh process whose "RCCL" is tests/stub_rccl.c
(PIE_RCCL_LIB), three shards of one corpus on GPU 0.  Checks, against the oracle's scan of the unsharded table:
  - the synchronous per-query exchange, incl. the capacity overflow that every rank sees in the gathered headers;
  - the pipelined union exchange (step_begin / finish / collect, two steps begun ahead, rotating buffer sets), 7 and 40 queries;
  - its overflow path (a union that outgrows the reservation is reported at collect, re-reserved, repeated).
usage: comm_stub_worker.py WORLD N_ROWS N_USERS"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
stub_dir = os.path.join(REPO, "tests", "_stub")
os.makedirs(stub_dir, exist_ok=True)
stub = os.path.join(stub_dir, "libstub_rccl.so")
subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-I/opt/rocm/include", "-o", stub, os.path.join(REPO, "tests", "stub_rccl.c"),
                "-L/opt/rocm/lib", "-lamdhip64"], check=True)
os.environ["PIE_RCCL_LIB"] = stub

import numpy as np
import torch  # noqa: F401  (before libpie_hip.so initialises HIP)
import oracle_py
import sph_pie_amd as pie

T0, DAY, SEED = 1700000000000, 86400 * 1000, 0x5EED5EED


def main():
    world, n, U = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    D = 32
    comm = pie.PieComm([0] * world)
    comm.gen_synthetic_sharded(SEED, n, U, D, 1)
    ctxs = [comm.ctx(r) for r in range(world)]
    maps = []
    for r in range(world):
        ctxs[r].set_disciplines(0xFFFFFFFF, D)
        rows_g, users_g = ctxs[r].shard_maps()
        maps.append((rows_g.astype(np.int64), users_g[: ctxs[r].n_users].astype(np.int64)))
    assert sum(m[0].size for m in maps) == n
    cols = oracle_py.gen(SEED, n, 0, n, U, D, 1)

    def queries_of(k):
        return [(T0 - 6 * 3600 * 1000 - 977 * q, T0 - (61 + q % 2) * DAY, (0x55555555, 0xAAAAAAAA, 0xFFFFFFFF)[q % 3]) for q in range(k)]

    checks = 0
    # ---- synchronous exchange of per-query lists; the first call starts from a 1024-row capacity and grows it from the headers
    qs = queries_of(7)
    want = [oracle_py.scan(*cols, U, *q) for q in qs]
    ms = comm.scan_batch_gather(qs)
    for at in range(world):
        for q in range(len(qs)):
            wc, wo, wi = want[q]
            total = 0
            for r in range(world):
                off, idx = comm.read_gathered(at, r, q)
                assert ms[r][q] == idx.size
                total += idx.size
                rows_r, users_r = maps[r]
                for lu in range(users_r.size):
                    gu = int(users_r[lu])
                    assert np.array_equal(rows_r[idx[off[lu]:off[lu + 1]]], wi[wo[gu]:wo[gu + 1]]), (at, q, r, gu)
            assert total == wi.size
            checks += 1

    def check_union(step, qs_, want_):
        nonlocal checks
        for at in range(world):
            got = [comm.step_read_gathered(at, r, step) for r in range(world)]
            for q in range(len(qs_)):
                wc, wo, wi = want_[q]
                total = 0
                for r in range(world):
                    uoff, rows, masks = got[r]
                    rows_r, users_r = maps[r]
                    sel = ((masks >> np.uint64(q)) & np.uint64(1)) == 1
                    csum = np.concatenate([[0], np.cumsum(sel)])
                    total += int(sel.sum())
                    for lu in range(users_r.size):
                        gu = int(users_r[lu])
                        a, b = int(uoff[lu]), int(uoff[lu + 1])
                        assert np.array_equal(rows_r[rows[a:b][sel[a:b]]], wi[wo[gu]:wo[gu + 1]]), (step, at, q, r, gu)
                    assert csum[-1] == sel.sum()
                assert total == wi.size
                checks += 1

    def run_pipelined(qs_, want_, k, first_step):
        """begin(i+1) | finish(i) | collect(i-1): two steps begun ahead of the exchange, as the header prescribes"""
        comm.step_begin(qs_)
        collected = []
        for i in range(k):
            if i + 1 < k:
                comm.step_begin(qs_)
            ms_ = comm.step_finish()
            assert [sum(ms_[r][q] for r in range(world)) for q in range(len(qs_))] == [int(w[2].size) for w in want_]
            if i >= 1:
                collected.append(comm.step_collect())
        collected.append(comm.step_collect())
        assert collected == list(range(first_step, first_step + k)), collected
        for st in collected[-3:]:            # the buffers of the last steps are still there (four rotating sets)
            check_union(st, qs_, want_)
        return first_step + k

    # ---- pipelined union exchange: first with a reservation that is too small (every rank learns it at collect) ...
    comm.step_reserve(len(qs), 0, 16)
    comm.step_begin(qs)
    comm.step_finish()
    try:
        comm.step_collect()
        raise AssertionError("a union of thousands of rows fitted 16?")
    except pie.PieError as ex:
        assert ex.code == pie.binding.PIE_E_CAPACITY, ex
    need = comm.needed_cap()
    assert need > 16
    comm.step_reserve(len(qs), 0, need)
    nxt = run_pipelined(qs, want, 9, 1)
    # ... then a batch of 40 queries (two mask words per union row): a new reservation, more steps
    qs40 = queries_of(40)
    want40 = [oracle_py.scan(*cols, U, *q) for q in qs40]
    comm.step_reserve(40, 0, need)
    nxt = run_pipelined(qs40, want40, 6, nxt)
    # the synchronous form still works between pipelined runs
    ms = comm.scan_batch_gather(qs[:3])
    assert [ms[r][0] for r in range(world)] == [int(comm.read_gathered(0, r, 0)[1].size) for r in range(world)]
    comm.close()
    print("comm stub ok: world %d, %d checks" % (world, checks))


if __name__ == "__main__":
    main()
