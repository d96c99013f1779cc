#!/bin/bash
# tools/prof_fz.sh "<counters>" flag flag ... — PMC counters of k_scan_stats under FQH_FZ_DBG knock-out flags
# (run on the GPU box via gpurun): exact instruction counts per part of the kernel.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_fz
rm -rf $OUT; mkdir -p $OUT
CTR=$1; shift
for f in "$@"; do
  FQH_FZ_DBG=$f timeout 300 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/f$f -o pmc -- python tools/exp_fzone.py 4 2 > $OUT/f$f.log 2>&1
  grep FQH_FZ $OUT/f$f.log
  python3 - "$OUT/f$f" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "k_scan_stats" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("   " + "  ".join("%s %.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc.items())))
PY
done
