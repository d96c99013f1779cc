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
            if "k_index" in k or "k_emit" in k or "k_stats" in k or "k_read" in k or "k_scan" in k:
                print("  ", k)
                for c, v in cs.items():
                    print("      %-28s n=%3d mean %.6g" % (c, len(v), sum(v) / len(v)))

# agreement check: rocprofv3's per-dispatch durations of the dominant kernel against the HIP-event
# measurement bench.py made in the same process (the timed steps are the last `steps` dispatches)
import json
tr = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)
log = os.path.join(out, "trace.log")
if tr and os.path.exists(log):
    lines = [l for l in open(log) if l.startswith("{")]
    if lines:
        b = json.loads(lines[-1])
        name = b["roofline"]["kernel"]
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
             for r in csv.DictReader(open(tr[0])) if name in r["Kernel_Name"]]
        k = b["steps"]
        if len(d) >= k:
            print("== agreement check (same process, bench.py under rocprofv3 --kernel-trace)")
            print("   %s per-dispatch ms: %s" % (name, " ".join("%.3f" % x for x in d)))
            print("   mean of the %d timed dispatches: %.4f ms; bench.py roofline.kernel_ms of that run (HIP events): %.4f ms"
                  % (k, sum(d[-k:]) / k, b["roofline"]["kernel_ms"]))
