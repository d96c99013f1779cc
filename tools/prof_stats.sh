#!/bin/bash
# rocprofv3 PMC passes focused on the histogram kernel (run on the GPU box via gpurun)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_stats
rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES" \
            "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ATOMIC_RETURN SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT" \
            "FETCH_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$tag -o pmc -- $CMD > $OUT/pmc_$tag.log 2>&1
done
python3 tools/prof_summary.py $OUT
