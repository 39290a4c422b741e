#!/usr/bin/env python3
"""Batch lanes (pie_set_batch_lanes) against the table size: a 64-query batch over a shard of cfg3 is one launch of ~20 us that
fills a fraction of the chip — latency, not bytes — so batches of different lanes (independent streams) run side by side.
For every table size (a 1/8, 1/4, 1/2 shard of cfg3 and the whole of it) and lane count: ms per batch of
PieScan.scan_batch_pipelined (three batches in flight per lane, ONE host thread), median of 5 regions.
usage: lanes_probe.py [Q] [sizes, comma separated rows]   (users = rows / 1000)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (before the first context: see DESIGN section 8)
import sph_pie_amd as pie

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SIZES = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [12_500_000, 25_000_000, 50_000_000, 100_000_000]
T0, DAY = 1700000000000, 86400000
now, cutoff, mask = T0 - 6 * 3600 * 1000, T0 - 61 * DAY, 0x55555555
qs = [(now - 977 * q, cutoff, mask) for q in range(Q)]

pie.build_hip()
ctx = pie.PieScan(0)


def timed(k, reps=5):
    out = []
    for _ in range(reps):
        ctx.synchronize()
        t = time.perf_counter()
        ctx.scan_batch_pipelined(k, qs)
        ctx.synchronize()
        out.append((time.perf_counter() - t) * 1e3 / k)
    out.sort()
    return out[len(out) // 2]


print("Q = %d; ms per batch, median of 5 regions; one host thread" % Q)
for n in SIZES:
    u = n // 1000
    ctx.gen_synthetic(0x5EED5EED, n, 0, n, u, 32, 0)
    ctx.set_disciplines(0xFFFFFFFF, 32)
    ctx.set_batch_lanes(0)
    auto = ctx.batch_lanes()
    row = []
    k = max(300, min(2400, int(3e10 / n)))
    for lanes in (1, 2, 3, 4):
        ctx.set_batch_lanes(lanes)
        ctx.scan_batch_pipelined(60, qs)
        row.append(timed(k))
    print("rows %11d users %7d: " % (n, u) + "  ".join("%d lane%s %.5f" % (i + 1, "s" if i else " ", v) for i, v in enumerate(row)) +
          "   (automatic: %d)" % auto, flush=True)
ctx.close()
