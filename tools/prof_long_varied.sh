#!/bin/bash
# tools/prof_long_varied.sh — kernel trace + PMC rows of the long-read route over bench.py's two legs (tools/exp_long_leg.py): reads of
# one length, and reads of log-normal length through the device plan (k_long_census / k_long_plan / k_long_lists / k_stats_long)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_long_varied
rm -rf $OUT; mkdir -p $OUT
CMD="python tools/exp_long_leg.py fastq-rs_amd/libfastq_hip.so"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
grep -E "k_long|k_stats_long|k_index_t|k_emit" $(find $OUT/trace -name "*kernel_stats.csv" | head -1) | sed 's/(.*)"/"/' | cut -d, -f1-7
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$tag -o pmc -- $CMD > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
# the launches of k_stats_long: the first four are the leg of one length, the last four the varied one
rows=collections.defaultdict(lambda: collections.defaultdict(float)); order=[]
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_stats_long(' not in r['Kernel_Name']: continue
    d=int(r['Dispatch_Id'])
    if d not in order: order.append(d)
    rows[d][r['Counter_Name']]+=float(r['Counter_Value'])
order.sort()
half=len(order)//2
for name,ids in (("one length",order[:half]),("log-normal",order[half:])):
    acc=collections.defaultdict(float)
    for d in ids:
        for c,v in rows[d].items(): acc[c]+=v/len(ids)
    print("k_stats_long,", name, {c: round(v,1) for c,v in acc.items()})
PY
done
