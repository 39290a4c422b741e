#!/usr/bin/env python3
"""Fold the rocprofv3 PMC passes of tools/run_pmc.sh into profiles/: per-kernel counter summaries plus
profiles/traffic.json (HBM bytes per launch of every table-pass kernel the command ran), which bench.py quotes as
roofline.traffic for the kernel it timed.

usage: tools/pmc_traffic.py <gpurun_out dir> <tag> [traffic file name, default traffic.json] [workload description]

gfx950 corrections (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE counts 128-B read requests at 64 B, so read bytes = 2 x FETCH_SIZE x 1024.  Cross-check kept beside it:
TCC_EA0_RDREQ_sum x 128 B (no 32-B requests observed).  Calibration point, re-measured with every run of this script:
the every-byte form (k_scan_compact<4, true, false, ...>, variant 0x01) must read the algorithmic 2.400 GB by both
formulas."""
import collections
import csv
import glob
import json
import os
import sys


def summarise(path):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void pie::", "").replace("pie::", "").strip()
        acc.setdefault((name, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    return acc


def main():
    root, tag = sys.argv[1], sys.argv[2]
    out_name = sys.argv[3] if len(sys.argv) > 3 else "traffic.json"
    workload = sys.argv[4] if len(sys.argv) > 4 else "bench.py default: 1e8 sessions / 1e5 users / 32 disciplines, random order, auth variant, spec query"
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    per = collections.OrderedDict()   # kernel -> counter -> (dispatches, mean)
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
                if k.startswith("k_scan_") or k.startswith("k_expired_stage") or k.startswith("k_ord_"):
                    per.setdefault(k, {})[c] = (len(v), sum(v) / len(v))
    stats = glob.glob(os.path.join(root, "%s_stats" % tag, "*", "*kernel_stats.csv"))
    avg_ns = {}
    if stats:
        with open(stats[0]) as f, open(os.path.join(out_dir, "%s_kernel_stats.csv" % tag), "w") as g:
            text = f.read()
            g.write(text)
        for r in csv.DictReader(open(stats[0])):
            avg_ns[r["Name"].split("(")[0].replace("void pie::", "").replace("pie::", "").strip()] = (int(r["Calls"]), float(r["AverageNs"]))
    kernels = collections.OrderedDict()
    for k, c in per.items():
        if "FETCH_SIZE" not in c:
            continue
        read_b = 2.0 * c["FETCH_SIZE"][1] * 1024.0
        write_b = c.get("WRITE_SIZE", (0, 0.0))[1] * 1024.0
        doc = {"dispatches": c["FETCH_SIZE"][0], "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b,
               "hbm_bytes_per_launch": read_b + write_b,
               "rdreq_x128_bytes_per_launch": c.get("TCC_EA0_RDREQ_sum", (0, 0.0))[1] * 128.0,
               "rdreq_32B": c.get("TCC_EA0_RDREQ_32B_sum", (0, 0.0))[1]}
        if k in avg_ns:
            doc["kernel_trace_calls"], doc["kernel_trace_avg_us"] = avg_ns[k][0], avg_ns[k][1] / 1e3
            doc["hbm_gbs_at_kernel_trace_avg"] = doc["hbm_bytes_per_launch"] / avg_ns[k][1]
            doc["frac_of_8TBs"] = doc["hbm_gbs_at_kernel_trace_avg"] / 8000.0
        kernels[k] = doc
    lib_hash = None
    try:   # the digest of the sources the profiled library was built from (sph-pie_amd/build.py writes it beside the .so)
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sph-pie_amd", "libpie_hip.so.srchash")) as f:
            lib_hash = f.read().strip()
    except OSError:
        pass
    out = {
        "lib_srchash": lib_hash,
        "source": "profiles/%s_{fetch,write,rdreq}.summary.csv + profiles/%s_kernel_stats.csv (tools/run_pmc.sh %s)" % (tag, tag, tag),
        "tag": tag,
        "workload": workload,
        "method": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_sum (separate passes, --kernel-trace only); read = 2 x "
                  "FETCH_SIZE KiB (gfx950 half-count of 128-B requests), write = WRITE_SIZE KiB; kernel time from a separate "
                  "--kernel-trace --stats run of the same command",
        "kernels": kernels,
    }
    with open(os.path.join(out_dir, out_name), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
