#!/bin/bash
# tools/exp_alloc_kind.sh [methods] — VERDICT r5 item 4 on the GPU box: (1) the index kernel's time per allocation METHOD,
# (2) PMC rows of fast and slow hipMalloc inputs: L2 <-> fabric request counts, latency level, credit / write stalls, and the
# UTCL1 (TLB) counters.  At most two TCC counters per pass (more: "exceeds the capabilities of the hardware", and rocprofv3
# then hangs — hence the short timeouts).  Output: gpurun_out/alloc/.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/alloc
rm -rf $OUT; mkdir -p $OUT
if [ "${1:-}" != "pmc-only" ]; then
  timeout 600 tools/bin/exp_alloc_kind ${1:-malloc,contig,vmm_one,malloc} 4 16 > $OUT/kinds.txt 2>&1
fi
i=0
for pass in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" \
            "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
            "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
            "TCC_TAG_STALL_sum TCC_LATENCY_FIFO_FULL_sum" \
            "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" \
            "TCC_EA0_RDREQ" \
            "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"; do
  i=$((i+1))
  FQH_EXP_PMC=1 timeout 90 rocprofv3 --pmc $pass --kernel-trace --output-format csv json -d $OUT/pmc_$i -o pmc -- tools/bin/exp_alloc_kind malloc 6 16 > $OUT/pmc_$i.log 2>&1
  echo "pass $i ($pass) rc $?" >> $OUT/passes.txt
done
python3 tools/alloc_kind_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -80
