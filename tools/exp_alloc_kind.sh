#!/bin/bash
# tools/exp_alloc_kind.sh — VERDICT r5 item 4 on the GPU box: (1) the index kernel's time per allocation METHOD, (2) PMC rows of
# fast and slow hipMalloc inputs (per-channel fabric reads, L2 requests, TLB counters).  Output: gpurun_out/alloc/.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/alloc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
timeout 900 tools/bin/exp_alloc_kind ${1:-malloc,contig,vmm_one,vmm1g,vmm2m,malloc} 4 16 > $OUT/kinds.txt 2>&1
i=0
for pass in "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_REQ TCC_HIT TCC_MISS" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
            "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" \
            "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"; do
  i=$((i+1))
  FQH_EXP_PMC=1 timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv json -d $OUT/pmc_$i -o pmc -- tools/bin/exp_alloc_kind malloc 6 16 > $OUT/pmc_$i.log 2>&1
  echo "pass $i rc $?" >> $OUT/passes.txt
done
ls -R $OUT | head -50
