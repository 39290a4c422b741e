#!/usr/bin/env python3
"""every-byte forms of the table pass (24 B/row really read), one scan at a time: k1 / whole-scan time per variant and grid
usage: python tools/sweep_full_read.py [grid ...]   (grid = PIE_K1_BLOCKS, read at table load; default: the library's own)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa
import sph_pie_amd as pie
T0 = 1700000000000; DAY = 86400000
now, cutoff = T0 - 6 * 3600 * 1000, T0 - 61 * DAY
grids = sys.argv[1:] or [""]
for grid in grids:
    if grid: os.environ["PIE_K1_BLOCKS"] = grid
    else: os.environ.pop("PIE_K1_BLOCKS", None)
    ctx = pie.PieScan(0)
    ctx.gen_synthetic(0x5EED5EED, 10 ** 8, 0, 10 ** 8, 10 ** 5, 32, 0)
    ctx.set_disciplines(0x5555555555555555, 32)
    ctx.set_profiling(1)
    for form in (0x01, 0x81, 0x21) if len(grids) > 1 else (0x01, 0x81, 0x21, 0x00, 0x80, 0x01):
        ctx.set_scan_form(form)
        for _ in range(5): ctx.scan_device(now, cutoff)
        ctx.stats_reset()
        for _ in range(30): ctx.scan_device(now, cutoff)
        st = ctx.stats()
        k1, sc = st["k1_ms_sum"] / st["n_profiled"], st["scan_ms_sum"] / st["n_profiled"]
        print("grid %6s form %#04x blocks %5d  k1 %.4f ms (%.3f of 8 TB/s)  t_scan %.4f ms (%.3f)" % (grid or "-", form, st["k1_blocks"], k1, 2.4 / k1 / 8, sc, 2.4 / sc / 8), flush=True)
    ctx.close()
