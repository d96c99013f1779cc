#!/bin/bash
# tools/prof_long.sh [L] — PMC rows of the long-read histogram kernel (k_stats_long) under tools/exp_longreads.py
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=${1:-5000}
OUT=gpurun_out/prof_long
rm -rf $OUT; mkdir -p $OUT
CMD="python tools/exp_longreads.py $L 8 x"
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_TA_BUSY_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-30)
  EXP_HIFI=1 timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$tag -o pmc -- $CMD > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0][-40:]
    if 'stats_long' not in k: continue
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    key=(k,r['Dispatch_Id']);
    if key not in seen: seen.add(key); n[k]+=1
for k in acc:
    print(k, n[k], {c: round(v/n[k],1) for c,v in acc[k].items()})
PY
done
