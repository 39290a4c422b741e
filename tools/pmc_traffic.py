#!/usr/bin/env python3
"""Fold the rocprofv3 PMC passes of tools/run_pmc.sh into profiles/: per-kernel counter summaries plus
profiles/k1_traffic.json (HBM bytes per launch of the scan kernel), which bench.py reports as roofline.traffic.

gfx950 corrections (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE counts 128-B read requests at 64 B, so read bytes = 2 x FETCH_SIZE x 1024.  Cross-check kept beside it:
TCC_EA0_RDREQ_sum x 128 B (no 32-B requests observed).  Calibration: the eager form (variant 0x01) reads exactly the
algorithmic 2.400 GB by both formulas (profiles/r01_pmc/pmc_fetch_0x01.summary.csv, pmc_rdreq_0x01.summary.csv)."""
import collections
import csv
import glob
import json
import os
import sys


def summarise(path):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        acc.setdefault((r["Kernel_Name"].split("(")[0], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    return acc


def main():
    root, tag = sys.argv[1], sys.argv[2]
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    vals = {}
    for kind in ("fetch", "write", "rdreq"):
        files = glob.glob(os.path.join(root, "%s_%s" % (tag, kind), "*", "*counter_collection.csv"))
        if not files:
            continue
        acc = summarise(files[0])
        with open(os.path.join(out_dir, "%s_%s.summary.csv" % (tag, kind)), "w") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "counter", "dispatches", "mean", "min", "max"])
            for (k, c), v in acc.items():
                w.writerow([k, c, len(v), sum(v) / len(v), min(v), max(v)])
                if "k_scan_" in k:
                    # the first dispatch of a table uses the streaming form and synchronous scans the stand-alone table
                    # pass; the steady-state launch is the table pass that carries the previous scan's K2 (with_tail):
                    # take that one when it ran, else the form with most dispatches
                    cur = vals.setdefault(c, (k, 0, 0.0))
                    better = ("with_tail" in k and "with_tail" not in cur[0]) or \
                             (("with_tail" in k) == ("with_tail" in cur[0]) and len(v) > cur[1])
                    if better:
                        vals[c] = (k, len(v), sum(v) / len(v))
    stats = glob.glob(os.path.join(root, "%s_stats" % tag, "*", "*kernel_stats.csv"))
    if stats:
        with open(stats[0]) as f, open(os.path.join(out_dir, "%s_kernel_stats.csv" % tag), "w") as g:
            g.write(f.read())
    if "FETCH_SIZE" in vals:
        kernel = vals["FETCH_SIZE"][0]
        read_b = 2.0 * vals["FETCH_SIZE"][2] * 1024.0
        write_b = vals.get("WRITE_SIZE", (None, 0, 0.0))[2] * 1024.0
        doc = {
            "kernel": kernel, "tag": tag, "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b,
            "hbm_bytes_per_launch": read_b + write_b,
            "rdreq_x128_bytes_per_launch": vals.get("TCC_EA0_RDREQ_sum", (None, 0, 0.0))[2] * 128.0,
            "rdreq_32B": vals.get("TCC_EA0_RDREQ_32B_sum", (None, 0, 0.0))[2],
            "workload": "bench.py default: 1e8 sessions / 1e5 users / 32 disciplines, random order, auth variant, spec query",
            "method": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_sum (separate passes); read = 2 x FETCH_SIZE KiB "
                      "(gfx950 half-count), write = WRITE_SIZE KiB",
        }
        with open(os.path.join(out_dir, "k1_traffic.json"), "w") as f:
            json.dump(doc, f, indent=1)
        print(json.dumps(doc))


if __name__ == "__main__":
    main()
