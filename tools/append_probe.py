#!/usr/bin/env python3
"""k_ord_append under rocprofv3: batches of 1000 in-order rows into tables of two sizes (run with rocprofv3 --kernel-trace --stats)"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa
import sph_pie_amd as pie
T0 = 1700000000000; DAY = 86400000; HOUR = 3600000
n, U = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1)
with pie.PieScan(0) as ctx:
    ctx.gen_synthetic(0x5EED5EED, n, 0, n, U, 32, 0)
    ctx.set_disciplines(2 ** 64 - 1, 32)
    ctx.set_ordered_run(2)
    now = T0 + HOUR
    ctx.append_rows(np.array([now], np.int64), np.array([now + HOUR], np.int64), np.array([0], np.int32), np.array([0], np.int32), U)   # capacity growth
    ctx.scan_device(now, T0 - 61 * DAY)
    for step in range(40):
        k = 1000
        s2 = np.sort(now + rng.integers(0, 1000, k)).astype(np.int64)
        now += 1000
        ctx.append_rows(s2, s2 + 12 * HOUR, rng.integers(0, U, k).astype(np.int32), rng.integers(0, 32, k).astype(np.int32), U)
    print(ctx.table_info()["ordered_rows"], ctx.table_info()["ordered_builds"])
