#!/usr/bin/env python3
"""Where does the time of pie_read_results go?  scan, then host copies of counts / offsets / idx, pinned vs pageable."""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import sph_pie_amd as pie

T0, DAY = 1700000000000, 86400 * 1000
N, U, D = 10 ** 8, 10 ** 5, 32
now, cutoff = T0 - 6 * 3600 * 1000, T0 - 61 * DAY
ctx = pie.PieScan(0)
ctx.gen_synthetic(0x5EED5EED, N, 0, N, U, D, 0)
ctx.set_disciplines(0x55555555, D)
for _ in range(3):
    m = ctx.scan_device(now, cutoff)
pc = torch.empty(U, dtype=torch.int32).pin_memory()
po = torch.empty(U + 1, dtype=torch.int64).pin_memory()
pi = torch.empty(m, dtype=torch.int32).pin_memory()
nc, no, ni = np.empty(U, np.int32), np.empty(U + 1, np.int64), np.empty(m, np.int32)


def t(label, fn, reps=20):
    fn()
    a = time.perf_counter()
    for _ in range(reps):
        fn()
    print("%-44s %.3f ms" % (label, (time.perf_counter() - a) * 1e3 / reps), flush=True)


t("scan only", lambda: ctx.scan_device(now, cutoff))
t("counts -> pinned", lambda: ctx.read_results_into(pc.data_ptr()))
t("offsets -> pinned", lambda: ctx.read_results_into(None, po.data_ptr()))
t("idx -> pinned", lambda: ctx.read_results_into(None, None, pi.data_ptr(), m))
t("counts+offsets -> pinned", lambda: ctx.read_results_into(pc.data_ptr(), po.data_ptr()))
t("all three -> pinned", lambda: ctx.read_results_into(pc.data_ptr(), po.data_ptr(), pi.data_ptr(), m))
t("all three -> pageable", lambda: ctx.read_results_into(nc.ctypes.data, no.ctypes.data, ni.ctypes.data, m))
t("scan + counts -> pinned", lambda: (ctx.scan_device(now, cutoff), ctx.read_results_into(pc.data_ptr())))
t("scan + all three -> pinned", lambda: (ctx.scan_device(now, cutoff), ctx.read_results_into(pc.data_ptr(), po.data_ptr(), pi.data_ptr(), m)))
t("scan + all three -> pageable", lambda: (ctx.scan_device(now, cutoff), ctx.read_results_into(nc.ctypes.data, no.ctypes.data, ni.ctypes.data, m)))
c, o, i = ctx.scan(now, cutoff)
t("ctx.scan() (numpy out, as the Node addon does)", lambda: ctx.scan(now, cutoff), 10)
t("scan + counts+offsets -> pinned", lambda: (ctx.scan_device(now, cutoff), ctx.read_results_into(pc.data_ptr(), po.data_ptr())))
t("scan + offsets -> pinned", lambda: (ctx.scan_device(now, cutoff), ctx.read_results_into(None, po.data_ptr())))
t("scan + idx -> pinned", lambda: (ctx.scan_device(now, cutoff), ctx.read_results_into(None, None, pi.data_ptr(), m)))
t("scan + all three -> pinned (again)", lambda: (ctx.scan_device(now, cutoff), ctx.read_results_into(pc.data_ptr(), po.data_ptr(), pi.data_ptr(), m)))
