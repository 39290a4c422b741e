#!/usr/bin/env python3
"""One-off scale check on the GPU box: a table far beyond the benchmark size (default 1e9 rows = 24 GB of columns,
~100 GB with the two result slots), generated on the device, scanned, and checked through size-independent
properties (the oracle cannot hold it)."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import sph_pie_amd as pie  # noqa: E402

T0, DAY = 1700000000000, 86400 * 1000
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 9
U, D = 10 ** 6, 32
now, cutoff, mask = T0 - 6 * 3600 * 1000, T0 - 61 * DAY, 0x55555555
with pie.PieScan(0) as ctx:
    t = time.time()
    ctx.gen_synthetic(0x5EED5EED, n, 0, n, U, D, 0)
    ctx.set_disciplines(mask, D)
    print("generated %d rows in %.2f s" % (n, time.time() - t), flush=True)
    for _ in range(3):
        ctx.scan_device(now, cutoff)
    ctx.set_profiling(1)
    ctx.stats_reset()
    m = ctx.scan_pipelined(10, now, cutoff)
    st = ctx.stats()
    k1 = st["k1_ms_sum"] / st["n_profiled"]
    print("M=%d  K1 %.3f ms = %.2f TB/s algorithmic (variant %s)" % (m, k1, 24.0 * n / k1 / 1e9, hex(st["k1_variant"])), flush=True)
    ctx.set_profiling(0)
    counts, offsets, idx = ctx.read_results()
    assert offsets[-1] == m == idx.size and np.array_equal(np.diff(offsets), counts)
    assert abs(m / n - 0.5 * 18 / (120 * 24)) < 1e-4
    assert np.unique(idx).size == m and idx.min() >= 0 and idx.max() < n
    s, e, u, d = ctx.fetch_rows(idx)
    assert np.all(e > now) and np.all(s >= cutoff) and np.all(((mask >> d.astype(np.uint64)) & 1) == 1)
    assert np.array_equal(u, np.repeat(np.arange(U, dtype=np.int32), counts))
    ok = (np.diff(s) > 0) | ((np.diff(s) == 0) & (np.diff(idx) > 0)) | (np.diff(u) != 0)
    assert np.all(ok)
    print("properties ok at n=%d" % n)
