#!/usr/bin/env python3
"""Summarises tools/prof.sh output: per-kernel average duration (kernel-trace stats) and PMC
counter sums per dispatch, averaged over the dispatches of each kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", f)
    for row in csv.DictReader(open(f)):
        print("  %-60s calls %5s avg %12s ns total %12s ns %6s%%" % (
            row.get("Name", "")[:60], row.get("Calls"), row.get("AverageNs"), row.get("TotalDurationNs"), row.get("Percentage")))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")[:40]
            acc[k][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
        print("== pmc:", os.path.basename(d))
        for k, cs in acc.items():
            if "k_index" in k or "k_emit" in k or "k_stats" in k or "k_read" in k:
                print("  ", k)
                for c, v in cs.items():
                    print("      %-28s n=%3d mean %.6g" % (c, len(v), sum(v) / len(v)))
