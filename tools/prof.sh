#!/bin/bash
# tools/prof.sh — rocprofv3 passes over a short bench run (run on the GPU box via gpurun).
# Output under gpurun_out/prof/; summaries are copied into profiles/ by hand.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
VAR=${FQH_INDEX_VARIANT:-0}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES" \
            "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$tag -o pmc -- $CMD > $OUT/pmc_$tag.log 2>&1
done
find $OUT -name "*.csv" | head -40
python3 tools/prof_summary.py $OUT
