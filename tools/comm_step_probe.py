#!/usr/bin/env python3
"""One-rank cost of a step of the exchange, three ways, on the same 10^8-row table ("the 1-rank C-ABI step
within 10 % of BatchedFeeds on the same shard"):
  scan only            ctx.scan_batch_pipelined (no exchange)
  BatchedFeeds         shard.py's Python driver, union message, over a 1-rank RCCL group (torch.distributed, nccl)
  pie_comm_step_*      the C-ABI communicator's pipelined union exchange (RCCL opened by the library itself)
usage: comm_step_probe.py [Q] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import sph_pie_amd as pie
from sph_pie_amd.shard import BatchedFeeds, HipShardBackend

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
T0, DAY = 1700000000000, 86400000
N, U, D = 10 ** 8, 10 ** 5, 32
now, cutoff, mask = T0 - 6 * 3600 * 1000, T0 - 61 * DAY, 0x55555555
qs = [(now - 977 * q, cutoff, mask) for q in range(Q)]


def timed(fn, reps=5):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t) * 1e3 / K)
    out.sort()
    return out[len(out) // 2]


torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

ctx = pie.PieScan(0)
ctx.gen_synthetic(0x5EED5EED, N, 0, N, U, D, 0)
ctx.set_disciplines(0xFFFFFFFF, D)
ctx.scan_batch_pipelined(20, qs)
scan_only = timed(lambda: (ctx.scan_batch_pipelined(K, qs), ctx.synchronize()))
res = {"Q": Q, "steps": K, "scan_only_ms": scan_only}
for g in (1, 8):
    bf = BatchedFeeds(HipShardBackend(ctx, dev), 0, 1, U, q_max=Q, always_collective=True, union=True, steps_per_gather=g)
    bf.run_steps(20, qs)
    res["batched_feeds_union_ms_g%d" % g] = timed(lambda: (bf.run_steps(K, qs), ctx.synchronize()))
ctx.close()

comm = pie.PieComm([0])
comm.gen_synthetic_sharded(0x5EED5EED, N, U, D, 0)
comm.ctx(0).set_disciplines(0xFFFFFFFF, D)
comm.step_reserve(Q, 0, 400000)


def comm_steps():
    comm.step_begin(qs)
    for i in range(K):
        if i + 1 < K:
            comm.step_begin(qs)
        comm.step_finish()
        if i >= 1:
            comm.step_collect()
    comm.step_collect()


comm_steps()
res["pie_comm_step_ms"] = timed(comm_steps)
res["ratio_comm_over_batched_feeds_g1"] = res["pie_comm_step_ms"] / res["batched_feeds_union_ms_g1"]
comm.close()
dist.destroy_process_group()
import json
print(json.dumps(res))
