#!/usr/bin/env python3
"""Every-byte scan (form 0x01 / 0x03) on cfg3: t_scan one at a time and the steady-state step with two in flight.
usage: full_read_probe.py [form]   (env PIE_STREAM_RIDE=0 for the two-launch form)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import sph_pie_amd as pie
form = int(sys.argv[1], 0) if len(sys.argv) > 1 else 1
T0, DAY = 1700000000000, 86400000
N, U = 10 ** 8, 10 ** 5
ctx = pie.PieScan(0)
ctx.gen_synthetic(0x5EED5EED, N, 0, N, U, 32, 0)
ctx.set_disciplines(0x55555555, 32)
now, cutoff = T0 - 6 * 3600 * 1000, T0 - 61 * DAY
ctx.set_scan_form(form)
for _ in range(5):
    ctx.scan_device(now, cutoff)
ctx.stats_reset(); ctx.set_profiling(1)
for _ in range(20):
    ctx.scan_device(now, cutoff)
st = ctx.stats(); ctx.set_profiling(0); ctx.stats_reset()
k1, ts = st["k1_ms_sum"] / st["n_profiled"], st["scan_ms_sum"] / st["n_profiled"]
print("form %#x: kernel %.4f ms (%.3f)  t_scan %.4f ms (%.3f)" % (form, k1, 2.4 / k1 / 8, ts, 2.4 / ts / 8))
res = {"0": [], "1": []}
for rep in range(6):           # alternate the two forms in ONE process: box-to-box and run-to-run spread cancels
    for ride in ("0", "1"):
        os.environ["PIE_STREAM_RIDE"] = ride
        ctx.scan_pipelined(10, now, cutoff)
        ctx.synchronize(); t = time.perf_counter(); ctx.scan_pipelined(200, now, cutoff); ctx.synchronize()
        res[ride].append((time.perf_counter() - t) * 5)
for ride in ("0", "1"):
    r = sorted(res[ride])
    print("  ride=%s step median %.4f ms (%.3f)  min %.4f  max %.4f" % (ride, r[len(r) // 2], 2.4 / r[len(r) // 2] / 8, r[0], r[-1]))
