"""The every-byte table pass (pie_set_scan_form 0x01) on cfg3 under queries that select 312 613 / 0 / 0 / 260 409 rows: what the selected rows (a returning histogram atomic + a scattered 16-byte slot store each) cost on top of the bytes (profiles/r03_zl_read_ceiling.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sph_pie_amd as pie
T0 = 1700000000000; DAY = 86400000
ctx = pie.PieScan(0)
ctx.gen_synthetic(0x5EED5EED, 10 ** 8, 0, 10 ** 8, 10 ** 5, 32, 0)
ctx.set_disciplines(0x5555555555555555, 32)
ctx.set_profiling(1)
ctx.set_scan_form(0x01)
for name, now, cutoff in (("spec query (312 613 selected)", T0 - 6 * 3600 * 1000, T0 - 61 * DAY), ("nothing live (0 selected)", T0 + 10 * DAY, T0 - 61 * DAY),
                          ("live but outside the window (0 selected)", T0 - 6 * 3600 * 1000, T0 + DAY), ("T0 - 3 h (half as many)", T0 - 3 * 3600 * 1000, T0 - 61 * DAY)):
    for _ in range(5): ctx.scan_device(now, cutoff)
    ctx.stats_reset()
    for _ in range(30): m = ctx.scan_device(now, cutoff)
    st = ctx.stats()
    k1, sc = st["k1_ms_sum"] / st["n_profiled"], st["scan_ms_sum"] / st["n_profiled"]
    print("%-45s M %8d  k1 %.4f ms (%.3f of 8 TB/s)  t_scan %.4f ms (%.3f)" % (name, m, k1, 2.4 / k1 / 8, sc, 2.4 / sc / 8), flush=True)
ctx.close()
