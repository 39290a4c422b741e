#!/usr/bin/env python3
"""Print a rocprofv3 --stats kernel summary: tools/kstats.py <output dir>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("%-52s calls=%-4s avg=%8.1f us  min=%8.1f" % (r["Name"][:52], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
