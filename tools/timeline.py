#!/usr/bin/env python3
"""Print a per-kernel timeline (us, relative) from a rocprofv3 --kernel-trace CSV: tools/timeline.py <csv> [first_row] [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:first + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f -> %9.1f  (%7.1f us)  q=%s stream=%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"),
                                                              r.get("Stream_Id", "?"), r["Kernel_Name"].split("(")[0][:40]))
