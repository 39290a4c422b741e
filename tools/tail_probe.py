#!/usr/bin/env python3
"""Time the batch's union tail kernel alone: N unpipelined batches (every finish launches k_union_tail by itself); run under
rocprofv3 --kernel-trace --stats.  usage: tail_probe.py [Q] [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import sph_pie_amd as pie
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
T0, DAY = 1700000000000, 86400000
ctx = pie.PieScan(0)
ctx.gen_synthetic(0x5EED5EED, 10 ** 8, 0, 10 ** 8, 10 ** 5, 32, 0)
ctx.set_disciplines(0xFFFFFFFF, 32)
qs = [(T0 - 6 * 3600 * 1000 - 977 * q, T0 - 61 * DAY, 0x55555555) for q in range(Q)]
for _ in range(N):
    ctx.scan_batch_begin(qs)
    ms = ctx.scan_batch_finish()
print(Q, ms[:3])
