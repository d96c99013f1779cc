#!/bin/bash
# tools/prof.sh — rocprofv3 passes over a short bench run (run on the GPU box via gpurun): a kernel trace with
# statistics, then PMC passes (each on its own: gpurun refuses --pmc together with the trace domains other than
# --kernel-trace).  Output under gpurun_out/prof/; tools/prof_summary.py turns it into profiles/roundN_*.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off --default-stream-gib 32 --default-shard-stream-gib 16"
PMC_CMD="$CMD --no-stream --no-shard-stream"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 400 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$tag -o pmc -- $PMC_CMD > $OUT/pmc_$tag.log 2>&1
done
# configs[3]: the streamed leg on its own, kernels AND memory copies (no counters in this pass)
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/stream -o stream -- python bench.py --stream-gib 32 --producer-threads 8 > $OUT/stream.log 2>&1
python3 tools/prof_summary.py $OUT ${1:-round6}
