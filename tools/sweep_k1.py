#!/usr/bin/env python3
"""Tuning sweep of the scan kernel on the GPU box: K1 variant x grid size, interleaved rounds in ONE process.
usage: python tools/sweep_k1.py [--rows N] [--variants 0x00,0x01,...] [--blocks 1536,3072,...] [--rounds R]"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import sph_pie_amd as pie  # noqa: E402

T0, DAY = 1700000000000, 86400 * 1000


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10 ** 8)
    ap.add_argument("--users", type=int, default=10 ** 5)
    ap.add_argument("--variants", default="0x00,0x01,0x02,0x03")
    ap.add_argument("--blocks", default="4096")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--query", default="spec")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--live-days", type=float, default=None, help="now = T0 - X days + 12 h, no window: live fraction ~ X/120")
    a = ap.parse_args()
    now, cutoff = (T0 - 6 * 3600 * 1000, T0 - 61 * DAY) if a.query == "spec" else (T0 - 100 * DAY, T0 - 61 * DAY)
    if a.query == "none":
        now, cutoff = 2 ** 62, -(2 ** 63)
    if a.query == "all":
        now, cutoff = -(2 ** 63), -(2 ** 63)
    if a.live_days is not None:
        now, cutoff = T0 - int(a.live_days * DAY) - 12 * 3600 * 1000, -(2 ** 63)
    mask = 0x55555555
    ref = None
    results = {}
    configs = [(int(v, 0), int(b)) for v in a.variants.split(",") for b in a.blocks.split(",")]
    for rnd in range(a.rounds):
        for v, b in configs:
            os.environ["PIE_K1_VARIANT"] = hex(v)
            os.environ["PIE_K1_BLOCKS"] = str(b)
            os.environ["PIE_K1_BLOCKS_LIVE"] = str(b)
            with pie.PieScan(0) as ctx:
                ctx.gen_synthetic(0x5EED5EED, a.rows, 0, a.rows, a.users, 32, a.flags)
                ctx.set_disciplines(mask, 32)
                for _ in range(3):
                    ctx.scan_device(now, cutoff)
                ctx.stats_reset()
                ctx.set_profiling(1)
                for _ in range(a.steps):
                    ctx.scan_device(now, cutoff)
                st = ctx.stats()
                ctx.set_profiling(0)
                if rnd == 0:
                    got = ctx.scan(now, cutoff)
                    if ref is None:
                        ref = got
                    else:
                        assert all(np.array_equal(x, y) for x, y in zip(got, ref)), "variant %x differs" % v
                k1 = st["k1_ms_sum"] / st["n_profiled"]
                sc = st["scan_ms_sum"] / st["n_profiled"]
                results.setdefault((v, b), []).append((k1, sc, st["k1_blocks"]))
    print("%8s %8s %8s | %9s %9s | %9s %9s | %8s" % ("variant", "blocks", "grid", "k1_med_ms", "k1_min_ms", "scan_med", "scan_min", "GB/s(k1)"))
    for (v, b), r in results.items():
        k1s = sorted(x[0] for x in r)
        scs = sorted(x[1] for x in r)
        print("%8s %8d %8d | %9.4f %9.4f | %9.4f %9.4f | %8.0f" % (hex(v), b, r[0][2], k1s[len(k1s) // 2], k1s[0], scs[len(scs) // 2], scs[0],
                                                                  24.0 * a.rows / (k1s[len(k1s) // 2] * 1e-3) / 1e9))


if __name__ == "__main__":
    main()
