"""tools/alloc_kind_summary.py DIR — per input of tools/exp_alloc_kind.sh's PMC passes: the index kernel's duration (trace
timestamps) next to the pass's counters, one row per k_index_fast launch; inputs 10 % slower than the pass's fastest are marked."""
import collections
import csv
import glob
import os
import sys

for d in sorted(glob.glob(os.path.join(sys.argv[1], "pmc_*"))):
    f = os.path.join(d, "pmc_counter_collection.csv")
    if not os.path.isfile(f):
        continue
    by = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "k_index_fast" not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        by.setdefault(k, collections.OrderedDict())
        by[k][r["Counter_Name"]] = by[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        by[k]["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if not by:
        continue
    best = min(v["ms"] for v in by.values())
    print("== %s" % d)
    for n, (k, v) in enumerate(by.items()):
        cols = "  ".join("%s %.4g" % (a, b) for a, b in v.items() if a != "ms")
        print("input %d launch %d: %.3f ms %s  %s" % (n // 2, n % 2, v["ms"], "SLOW" if v["ms"] > 1.04 * best else "    ", cols))
