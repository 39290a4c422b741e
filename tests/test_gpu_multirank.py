"""More than one rank with the HIP backend.  A GPU box has one MI355X and RCCL refuses two ranks on one
device, so the ranks are fresh child processes that share GPU 0 and exchange over gloo with pinned-host staging
(shard.py transport="host" / bench.py --transport gloo): the rehearsal of `bench.py --gpus N` end to end except the RCCL
transport itself, which the 1-rank RCCL tests of test_gpu_parity.py cover."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _env(rank, world, port):
    env = dict(os.environ)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(port), "PIE_BENCH_DEVICE": "0", "PIE_BENCH_TRANSPORT": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    return env


def _run_ranks(cmds, envs, timeout):
    procs = [subprocess.Popen(c, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=REPO) for c, e in zip(cmds, envs)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:   # exactly the children started here
            if p.poll() is None:
                p.kill()
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, "rank failed (rc %s):\n%s" % (p.returncode, se[-3000:])
    return outs


@pytest.mark.parametrize("world,n,U", [(2, 300_000, 1500), (3, 120_000, 401)])
def test_two_ranks_one_gpu_shard_and_exchange(pie, oracle, tmp_path, world, n, U):
    """pie_shard_table on ONE corpus in every rank, BatchedFeeds (union + lists, 1 / 3 / 8 steps per all-gather) and
    ShardedFeeds over HipShardBackend: the global feeds every rank rebuilds equal the oracle's on the unsharded table."""
    port = 30500 + (os.getpid() % 2000) + world
    worker = os.path.join(REPO, "tests", "multirank_worker.py")
    cmds = [[sys.executable, worker, str(r), str(world), str(port), str(n), str(U), str(tmp_path)] for r in range(world)]
    _run_ranks(cmds, [_env(r, world, port) for r in range(world)], 600)
    total_rows = 0
    for r in range(world):
        rep = json.load(open(tmp_path / ("rank%d.json" % r)))
        assert rep.get("ok") and rep["checks"] >= 40, rep
        total_rows += rep["rows_local"]
    assert total_rows == n


@pytest.mark.parametrize("extra", [[], ["--exchange", "lists"], ["--queries-per-launch", "1"], ["--queries-per-launch", "64"]])
def test_bench_two_ranks_one_gpu(pie, extra):
    """`bench.py --gpus 2` itself, two ranks on GPU 0 over the gloo transport: strong scaling (one corpus sharded by user hash),
    every timed region, gather_verified and the scan-only leg; rank 0 prints one well-formed line."""
    port = 32500 + (os.getpid() % 2000) + len(extra) + (7 if "64" in extra else 0)
    base = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--rows", "2000000", "--users", "3000", "--steps", "6", "--warmup", "2",
            "--repeat", "2", "--gather-batch", "4", "--no-cpu-baseline"] + extra
    outs = _run_ranks([base, base], [_env(r, 2, port) for r in range(2)], 600)
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["gather_verified"] is True
    assert line["config"]["sessions_total"] == 2000000 and line["config"]["users_total"] == 3000
    assert line["value"] > 0 and line["scan_only_ms_per_step"]["median"] > 0
    if not extra:
        assert line["exchange"]["format"] == "union"


@pytest.mark.parametrize("world,n,U", [(3, 200_000, 901), (2, 60_000, 77)])
def test_c_abi_communicator_with_several_ranks(pie, oracle, world, n, U):
    """pie_comm_* with world > 1 (ADVICE r02 / VERDICT r02 item 5): a fresh process whose RCCL is the stand-in of
    tests/stub_rccl.c drives `world` shards on GPU 0 — synchronous lists incl. the overflow every rank sees, the pipelined
    union exchange (begin / finish / collect), its overflow path, 7 and 40 queries; global feeds against the oracle."""
    res = subprocess.run([sys.executable, os.path.join(REPO, "tests", "comm_stub_worker.py"), str(world), str(n), str(U)],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=REPO)
    assert res.returncode == 0, res.stderr[-3000:]
    assert "comm stub ok" in res.stdout
