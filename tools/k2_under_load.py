#!/usr/bin/env python3
"""How long does a latency-bound K2-sized job take while a table pass saturates the memory system?  Context B scans a
tiny table over 10^5 users (its K1 is trivial, its K2 is the real one); context A runs the cfg3 scan loop in another
thread.  Compares B's time per scan alone and under A's load."""
import os, sys, threading, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import sph_pie_amd as pie

T0, DAY = 1700000000000, 86400 * 1000
now, cutoff = T0 - 6 * 3600 * 1000, T0 - 61 * DAY
A = pie.PieScan(0)
A.gen_synthetic(0x5EED5EED, 10 ** 8, 0, 10 ** 8, 10 ** 5, 32, 0)
A.set_disciplines(0x55555555, 32)
B = pie.PieScan(0)
nb = 400000   # every row live at `now_b`: 4 rows per user, K1 over 400k rows is a few microseconds
B.gen_synthetic(0x5EED5EED, nb, 0, nb, 10 ** 5, 32, 0)
B.set_disciplines(0xFFFFFFFF, 32)
now_b = T0 - 130 * DAY
for _ in range(5):
    A.scan_device(now, cutoff); B.scan_device(now_b, -(2 ** 63))


def loop_b(reps):
    t = time.perf_counter()
    B.scan_pipelined(reps, now_b, -(2 ** 63))
    B.synchronize()
    return (time.perf_counter() - t) * 1e3 / reps


print("B alone: %.4f ms/scan (variant %s)" % (loop_b(2000), hex(B.stats()["k1_variant"])), flush=True)
stop = False
done = [0]


def loop_a():
    while not stop:
        A.scan_pipelined(200, now, cutoff)
        done[0] += 200


th = threading.Thread(target=loop_a)
t0 = time.perf_counter()
th.start()
time.sleep(0.05)
b = loop_b(2000)
stop = True
th.join()
dt = time.perf_counter() - t0
print("B under A's load: %.4f ms/scan; A meanwhile: %.4f ms/scan" % (b, dt * 1e3 / max(done[0], 1)), flush=True)
